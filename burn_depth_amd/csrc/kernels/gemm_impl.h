// MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]          (both operands K-contiguous)
//
// Design (MI355X-first; see DESIGN.md "K4/K8"):
//  * 64-lane waves, 16x16 MFMA tiles in every kernel: v_mfma_f32_16x16x32_{bf16,f16}, 4 x v_mfma_f32_16x16x4_f32 (fp32
//    mode) or the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands), fp32 accumulation in all of them.
//    The MFMA "A" operand is the WEIGHT tile and the "B" operand the
//    activation tile, so a lane ends up holding 4 consecutive output columns n of ONE row m:
//    the epilogue stores 8-byte (bf16) / 16-byte (f32) vectors and per-column vectors (bias,
//    LayerScale) are loaded as float4.
//  * LDS tiles are rows of exactly 128 bytes (64 bf16 or 32 f32 of K) -- byte-identical addressing
//    for both precisions.  Tiles are filled with `global_load_lds_dwordx4` (no VGPR round trip);
//    the bank-conflict swizzle (16-byte chunk index XOR ((row>>1)&7)) is applied on the per-lane
//    GLOBAL source address and again on the ds_read_b128 address (the LDS image written by one
//    wave-instruction must stay lane-linear).
//  * gemm_kernel (128x128 / 256x32 tiles): two LDS stages; the global->LDS loads of k-tile t+1 are in flight while
//    k-tile t is multiplied; one barrier per k-tile, two workgroups per CU. gemm256_kernel (256x256): a ring of five
//    32-KB half-tile slots and the staggered two-group schedule described at its definition.
//  * A-operand modes: dense rows, indexed rows (token->map "merge" gather, encoder.rs:234-319) and
//    3x3 stride-1 pad-1 convolution taps over an NHWC tensor (out-of-image taps read a zero page).
//  * Grouped weights: up to 4 row ranges, each with its own W / bias / scale (the three ViT-L
//    encoders of Depth Pro advance layer by layer in ONE launch).
//  * XCD-aware block order: logical tile ids are dealt so that each XCD (private L2) works on a
//    contiguous range of tiles that share the same A row-panel.
#pragma once

#include <atomic>
#include <type_traits>
#include "elem.h"
#include "gemm.h"

namespace md {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

#define MD_SEL_G(arr, g) ((g) == 0 ? (arr)[0] : (g) == 1 ? (arr)[1] : (g) == 2 ? (arr)[2] : (arr)[3])

// 16x16 output tiles: v_mfma_f32_16x16x32_bf16 / 4 x v_mfma_f32_16x16x4_f32 per 16-byte operand pair.
// Same FLOPs per LDS byte as the 32x32 forms; the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS give-back item 7).
typedef __attribute__((ext_vector_type(4))) float f32x4acc_t;
template <typename T>
struct Atom16;
template <>
struct Atom16<bf16_t> {
  static __device__ __forceinline__ void mma(const i32x4_t& a, const i32x4_t& b, f32x4acc_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                0, 0, 0);
  }
};
template <>
struct Atom16<f16_t> {
  static __device__ __forceinline__ void mma(const i32x4_t& a, const i32x4_t& b, f32x4acc_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <>
struct Atom16<f16s_t> : Atom16<f16_t> {};  // split-half planes are IEEE halves: the same instruction on each plane
template <>
struct Atom16<float> {
  static __device__ __forceinline__ void mma(const i32x4_t& a, const i32x4_t& b, f32x4acc_t& c) {
    f32x4_t af = __builtin_bit_cast(f32x4_t, a), bf = __builtin_bit_cast(f32x4_t, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
  }
};

// fp8 (OCP e4m3) operands: a 16-byte fragment holds 16 k-values per lane = two 8-byte MFMA operands. Which k
// each lane holds does not matter as long as A and B agree (both are read with the same fragment address), so
// the LDS image, swizzle and fragment reads are byte-identical to the bf16 / f32 forms; only KE doubles to 128.
typedef long i64x2_t __attribute__((ext_vector_type(2)));
template <>
struct Atom16<fp8_t> {
  static __device__ __forceinline__ void mma(const i32x4_t& a, const i32x4_t& b, f32x4acc_t& c) {
    const i64x2_t av = __builtin_bit_cast(i64x2_t, a), bv = __builtin_bit_cast(i64x2_t, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av[0], bv[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av[1], bv[1], c, 0, 0, 0);
  }
};
// Block-scaled (MX) e4m3 MFMA, `v_mfma_scale_f32_16x16x128_f8f6f4`: 32 operand bytes per lane and instruction,
// twice the FLOPs per clock of the non-scaled fp8 / bf16 forms (MI355X_MICROARCH.md, Matrix cores). Every block scale is
// the E8M0 byte 127 = 2^0, so the products are the plain e4m3 products and the result equals the non-scaled forms' up to
// fp32 summation order. A lane's 32 bytes are two 16-byte LDS fragments (k sub-steps s and s+1, read at the same offsets
// for both operands): which k a lane holds is free as long as A and B agree.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
constexpr int kMxUnitScale = 0x7f7f7f7f;
__device__ __forceinline__ i32x8_t mx_cat(const i32x4_t& lo, const i32x4_t& hi) {
  return (i32x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void mx_mma16(const i32x4_t& a0, const i32x4_t& a1, const i32x4_t& b0, const i32x4_t& b1, f32x4acc_t& c) {
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mx_cat(a0, a1), mx_cat(b0, b1), c, 0, 0, 0, kMxUnitScale, 0, kMxUnitScale);
}
// element type of everything the epilogue reads / writes as "T" (outputs, residuals): bf16 for fp8 operands
template <typename T>
struct OutT {
  typedef T type;
};
template <>
struct OutT<fp8_t> {
  typedef bf16_t type;
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// exact-erf GELU (burn/DINOv2 `Gelu`). fp32 mode: libm erff. bf16 mode: erf(x/sqrt2) ~ x*P(x^2), a degree-8
// minimax polynomial on |x| <= 4.4 (clamped to +-1 outside), evaluated two elements per instruction with
// the packed fp32 VALU ops (v_pk_fma_f32): no transcendental, ~8 issue slots per element instead of ~26.
// |GELU abs err| <= 8.5e-5 over the whole real line -- below half a bf16 ulp of any |output| >= 0.05 --
// which matters because the 16-lane SIMDs make the fc1 epilogue (87 M GELUs per launch) VALU-bound.
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// (round 6) GELU(x) = x Phi(x), Phi = 1/2 + 1/2 erf(x / sqrt 2) = clamp01(1/2 + x P'(x^2)) with P' = P / 2 (the halved coefficients are exact):
// the [0, 1] clamp is the output modifier of the packed FMA that forms Phi (VOP3P `clamp`; hipcc does not fold a clamp into a packed fp32
// instruction by itself, hence the asm), and it also stands in for the range clamp of x: beyond the fitted |x| <= 4.4 the polynomial x P(x^2) is
// monotone and >= 1.00005 in magnitude (it runs off to +-inf with the sign of x: leading coefficient > 0), so Phi saturates to exactly 0 / 1 as the
// former med3(x, +-4.4) -> med3(e, +-1) pair made it. 11 packed instructions per element PAIR instead of 12 + 4 scalar med3; four elements are
// evaluated together so that no packed result is read by the very next instruction (gfx950's one-wait-state forwarding hazard behind VOP3P: hipcc
// fills it with s_nop otherwise, one per Horner step).
__device__ __forceinline__ f32x4_t gelu4_poly(f32x4_t x) {
  const f32x4_t t = x * x;
  f32x4_t p = __builtin_elementwise_fma((f32x4_t){4.414737672653324e-11f, 4.414737672653324e-11f, 4.414737672653324e-11f, 4.414737672653324e-11f}, t,
                   (f32x4_t){-4.49072912189763e-09f, -4.49072912189763e-09f, -4.49072912189763e-09f, -4.49072912189763e-09f});
#define MD_G8(c) p = __builtin_elementwise_fma(p, t, (f32x4_t){c, c, c, c})
  MD_G8(2.0082663354514807e-07f); MD_G8(-5.2469567890511825e-06f); MD_G8(9.019154822453856e-05f); MD_G8(-0.001093542668968439f);
  MD_G8(0.00978023186326027f); MD_G8(-0.06630592793226242f); MD_G8(0.39888665080070496f);
#undef MD_G8
  f32x2_t ph0, ph1;
  const f32x2_t x0 = {x[0], x[1]}, x1 = {x[2], x[3]}, p0 = {p[0], p[1]}, p1 = {p[2], p[3]};
  asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(ph0) : "v"(x0), "v"(p0));
  asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(ph1) : "v"(x1), "v"(p1));
  const f32x2_t o0 = x0 * ph0, o1 = x1 * ph1;
  return (f32x4_t){o0[0], o0[1], o1[0], o1[1]};
}

// split-half f16 mode (f16x2; the one-plane f16 mode until round 6): erf by Abramowitz & Stegun 7.1.26 (|erf error| <= 1.5e-7): one v_rcp_f32, one v_exp_f32 and ten plain VALU
// operations per element, about a third of libm's erff. GELU(x) = h + |h| - |h| * poly(t) * exp(-x^2/2), h = x/2,
// t = 1 / (1 + p |x| / sqrt 2): |abs error| <= 3.4e-7 and <= 1.7e-4 relative wherever |GELU| >= 1e-3, inside half an f16
// ulp (2.4e-4); the bf16 polynomial's 8.5e-5 absolute is not.
__device__ __forceinline__ float gelu_as(float x) {
  const float az = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(az * az * -1.4426950408889634f);
  const float h = 0.5f * x, ha = fabsf(h);
  return fmaf(-ha, p * e, h + ha);
}

// f16 mode, one plane (round 6; the split-half mode keeps gelu_as: 65.6 -> 65.3 ms there, and its documented error budget): Phi(x) = clamp01(1/2 + x Q'(w)), w = x^2 / 3.92^2 - 1 in [-1, 1] for |x| <= 5.54, Q' = Q / 2 of degree 16 in the
// SHIFTED variable (a monomial Horner form in x^2 itself loses 1e-3 in fp32 near |x| = 5), four elements per step on the packed fp32 pipe, no
// transcendental, the clamp as the packed FMA's output modifier (as gelu4_poly); |GELU abs error| <= 6.6e-7, <= 1.9e-4 relative where |GELU| >= 1e-3; the
// leading coefficient is positive, so beyond the fitted range x Q runs off to +-inf with the sign of x and the clamp finishes it.
__device__ __forceinline__ f32x4_t gelu4_poly16(f32x4_t x) {
  const f32x4_t u = x * x;
  const f32x4_t w = __builtin_elementwise_fma(u, (f32x4_t){0.0650770515203476f, 0.0650770515203476f, 0.0650770515203476f, 0.0650770515203476f},
                                              (f32x4_t){-1.0f, -1.0f, -1.0f, -1.0f});
  f32x4_t p = {1.114702245e-04f, 1.114702245e-04f, 1.114702245e-04f, 1.114702245e-04f};
#define MD_G16(c) p = __builtin_elementwise_fma(p, w, (f32x4_t){c, c, c, c})
  MD_G16(-3.182258806e-04f); MD_G16(2.995073155e-04f); MD_G16(-4.180655232e-04f); MD_G16(1.390643185e-03f);
  MD_G16(-2.821136033e-03f); MD_G16(4.429543856e-03f); MD_G16(-6.983477620e-03f); MD_G16(1.077367551e-02f);
  MD_G16(-1.539201476e-02f); MD_G16(2.057781629e-02f); MD_G16(-2.616278827e-02f); MD_G16(3.203918040e-02f);
  MD_G16(-3.860133142e-02f); MD_G16(4.740566761e-02f); MD_G16(-6.367799640e-02f); MD_G16(1.275397241e-01f);
#undef MD_G16
  f32x2_t ph0, ph1;
  const f32x2_t x0 = {x[0], x[1]}, x1 = {x[2], x[3]}, p0 = {p[0], p[1]}, p1 = {p[2], p[3]};
  asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(ph0) : "v"(x0), "v"(p0));
  asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(ph1) : "v"(x1), "v"(p1));
  const f32x2_t o0 = x0 * ph0, o1 = x1 * ph1;
  return (f32x4_t){o0[0], o0[1], o1[0], o1[1]};
}

template <typename T>
__device__ __forceinline__ f32x4_t gelu4(f32x4_t v) {
  if constexpr (std::is_same<T, float>::value) {  // fp32 parity mode: libm erff
    f32x4_t r = {gelu_erf<T>(v[0]), gelu_erf<T>(v[1]), gelu_erf<T>(v[2]), gelu_erf<T>(v[3])};
    return r;
  } else if constexpr (std::is_same<T, f16_t>::value) {
    // (the packed degree-16 polynomial against gelu_as, profiles/r06_gelu16_ab.txt: fc1 39.3 -> 36.9 ms per step in the f16 mode; round 5's form of it --
    // two med3 per pair, a hazard nop per Horner step -- had reached 39.3 from 40.3)
    return gelu4_poly16(v);
  } else if constexpr (is_half<T>::value) {
    f32x4_t r = {gelu_as(v[0]), gelu_as(v[1]), gelu_as(v[2]), gelu_as(v[3])};
    return r;
  } else {
    return gelu4_poly(v);
  }
}

__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  const unsigned q = __umulhi((unsigned)n, f.mul) >> f.sh;
  return f.d == 1 ? n : (int)q;
}

__device__ __forceinline__ f32x4_t relu4(f32x4_t v) {
  f32x4_t r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  return r;
}

// ------------------------------------------------------------------------------------------------
// Epilogue for 4 consecutive columns n..n+3 of row m (all bounds already checked by the caller).
// ------------------------------------------------------------------------------------------------
// x + s * v with ONE rounding per element, spelled as an fma so that every kernel of the family (the 256^2 kernel's
// read-modify-write epilogue and the per-vector epilogue of the small tiles) produces the same bits for the same operands:
// the tile choice -- and with it a tile-parallel split of the sequence range -- must not change the residual stream.
__device__ __forceinline__ f32x4_t resid_ls4(f32x4_t x, f32x4_t s, f32x4_t v) {
  return (f32x4_t){__builtin_fmaf(s[0], v[0], x[0]), __builtin_fmaf(s[1], v[1], x[1]), __builtin_fmaf(s[2], v[2], x[2]),
                   __builtin_fmaf(s[3], v[3], x[3])};
}
// a * b + c, one rounding per element (the same reason: q = acc * qscale + bias * qscale in every kernel of the family)
__device__ __forceinline__ f32x4_t fma4(f32x4_t a, f32x4_t b, f32x4_t c) {
  return (f32x4_t){__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1]), __builtin_fmaf(a[2], b[2], c[2]),
                   __builtin_fmaf(a[3], b[3], c[3])};
}

template <typename T>
__device__ __forceinline__ void epilogue4(const GemmParams& p, int g, int m, int n, f32x4_t v, long boff) {
  const float* bias = MD_SEL_G(p.bias, g);
  if (const float* ws = MD_SEL_G(p.wscale, g)) v *= *(const f32x4_t*)(ws + n) * p.ascale;  // fp8 operands: dequantise
  switch (p.epi) {
    case EPI_STORE: {
      if (bias) v += *(const f32x4_t*)(bias + n);
      const long rm = p.res_mod > 0 ? (long)(m - fdiv(m, p.fd_res_mod) * p.res_mod) : (long)m;
      if (p.res1) v += load4p<T>((const T*)p.res1 + rm * p.ldr + n, p.r_plane);
      if (p.res2) v += load4p<T>((const T*)p.res2 + (long)m * p.ldr + n, p.r_plane);  // res_mod wraps res1 only
      if (p.act == ACT_RELU) {
        v = relu4(v);
      } else if (p.act == ACT_GELU) {
        v = gelu4<T>(v);
      }
      if (p.out_f32)
        store4<float>((float*)p.out + boff + (long)m * p.ldo + n, v);
      else
        store4p<T>((T*)p.out + boff + (long)m * p.ldo + n, p.o_plane, v);
      if (p.out2) store4p<T>((T*)p.out2 + boff + (long)m * p.ldo + n, p.o_plane, relu4(v));
      break;
    }
    case EPI_RESID_LS: {
      const float* sc = MD_SEL_G(p.scale, g);
      float* x = (float*)p.out + (long)m * p.ldo + n;
      f32x4_t r = p.resid_src ? *(const f32x4_t*)(p.resid_src + (long)m * p.ldo + n) : *(const f32x4_t*)x;
      v += *(const f32x4_t*)(bias + n);
      f32x4_t s = *(const f32x4_t*)(sc + n);
      *(f32x4_t*)x = resid_ls4(r, s, v);
      break;
    }
    case EPI_PATCH_EMBED: {
      const float* pos = MD_SEL_G(p.pos, g);
      int t = fdiv(m, p.fd_seq_patches);
      int pi = m - t * p.seq_patches;
      long row = (long)t * p.seq_stride + 1 + pi;
      v += *(const f32x4_t*)(bias + n);
      v += *(const f32x4_t*)(pos + (long)(1 + pi) * p.embed + n);
      *(f32x4_t*)((float*)p.out + row * p.ldo + n) = v;
      break;
    }
    case EPI_QKV: {
      // q columns: acc * qscale + bias * qscale (the 256^2 kernel folds the scale into its bias / scale vectors; the same
      // two roundings here keep q independent of the tile). embed % 4 == 0: a 4-column vector never straddles q | k
      if (n < p.embed) {
        const f32x4_t qs = {p.qscale, p.qscale, p.qscale, p.qscale};
        v = fma4(v, qs, *(const f32x4_t*)(bias + n) * p.qscale);
      } else {
        v += *(const f32x4_t*)(bias + n);
      }
      const int two_d = 2 * p.embed;
      if constexpr (is_split<T>::value) {
        // split-half rows [q_hi | q_lo | k_hi | k_lo], each `embed` wide; V^T lo plane v_plane elements behind the hi plane
        if (n < two_d) {
          const int isk = n >= p.embed ? 1 : 0;
          store4s<T>((T*)p.out + (long)m * (2L * two_d) + isk * two_d + (n - isk * p.embed), p.embed, v);
        } else {
          const int c = n - two_d;
          const int hd = c >> 6, d = c & 63;
          const int seq = fdiv(m, p.fd_seq_stride);
          const int i = m - seq * p.seq_stride;
          T* vt = (T*)p.vT + (((long)seq * p.heads + hd) * 64 + d) * p.kpad + i;
          store1s<T>(vt, p.v_plane, v[0]);
          store1s<T>(vt + p.kpad, p.v_plane, v[1]);
          store1s<T>(vt + 2L * p.kpad, p.v_plane, v[2]);
          store1s<T>(vt + 3L * p.kpad, p.v_plane, v[3]);
        }
      } else if (n < two_d) {
        store4<T>((T*)p.out + (long)m * two_d + n, v);
      } else {
        int c = n - two_d;
        int hd = c >> 6, d = c & 63;
        int seq = fdiv(m, p.fd_seq_stride);
        int i = m - seq * p.seq_stride;
        long base = (((long)seq * p.heads + hd) * 64 + d) * p.kpad + i;
        T* vt = (T*)p.vT;
        store1<T>(vt + base, v[0]);
        store1<T>(vt + base + p.kpad, v[1]);
        store1<T>(vt + base + 2L * p.kpad, v[2]);
        store1<T>(vt + base + 3L * p.kpad, v[3]);
      }
      break;
    }
    case EPI_PIXSHUF: {
      int tap = fdiv(n, p.fd_psC);
      int co = n - tap * p.psC;
      const int f = p.ps_f;
      int dy = f == 4 ? tap >> 2 : tap >> 1, dx = tap - dy * f;
      int t = fdiv(m, p.fd_psW);
      int x = m - t * p.psW;
      int b = fdiv(t, p.fd_psH);
      int y = t - b * p.psH;
      long orow = ((long)b * f * p.psH + f * y + dy) * ((long)f * p.psW) + f * x + dx;
      if (bias) v += *(const f32x4_t*)(bias + co);
      long o = orow * p.ldo + p.ps_coff + co;
      if (p.out_f32)
        store4<float>((float*)p.out + o, v);
      else
        store4p<T>((T*)p.out + o, p.o_plane, v);
      if (p.out2) store4p<T>((T*)p.out2 + o, p.o_plane, relu4(v));
      break;
    }
    default:
      break;
  }
}

// ------------------------------------------------------------------------------------------------
// KSPLIT > 1 (launches that leave most CUs idle behind a long contraction: DA3 `small`'s fc2, K = 1536 on 132 workgroups; its stride-2
// stage convolution, K = 3456 on 36): KSPLIT groups of WGM x WGN waves share the output tile, group g multiplies the g-th part of the
// k-tiles through its own two-stage ring, the partial accumulators meet in LDS in a fixed order (group 0 + group 1 + ...) and group 0
// runs the epilogue -- the dependent k-loop is KSPLIT times shorter, the result is deterministic, and it differs from the unsplit
// kernel's in the last bits (another summation order of the same products).
template <typename T, int BM, int BN, int WGM, int WGN, int AMODE, int KSPLIT = 1>
__global__ __launch_bounds__(WGM* WGN * 64 * KSPLIT) void gemm_kernel(const GemmParams p) {
  constexpr int NW = WGM * WGN;  // waves of one k-group
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 16, TN = WTN / 16;  // 16x16 MFMA tiles per wave tile
  constexpr int RG_A = BM / 8, RG_W = BN / 8;  // 8-row groups (one glds wave-instruction each)
  static_assert(RG_A % NW == 0, "A row groups must divide evenly over the waves");
  constexpr int A_ITERS = RG_A / NW;
  constexpr int W_ITERS = (RG_W + NW - 1) / NW;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  typedef typename OutT<T>::type TO;  // element type of outputs / residuals
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KE = 128 / ESZ;  // K elements per 128-byte LDS row

  extern __shared__ __attribute__((aligned(16))) char smem_all[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KSPLIT > 1 ? wave_all / NW : 0;       // k-group
  const int wave = KSPLIT > 1 ? wave_all % NW : wave_all;
  const int wm = wave / WGN, wn = wave % WGN;
  char* const smem = smem_all + grp * (2 * STAGE_BYTES);  // this group's two-stage ring

  // ---- XCD-aware, bijective block -> logical tile id (blocks b and b+8 share an XCD) ----
  const int nwg = gridDim.x;
  int id;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // L2-aware raster: n-tiles are walked in groups of `gn` (host-chosen, <= 4): the 32 blocks an XCD
  // runs together cover (32/gn) A panels x gn W panels, and a W group (gn x 256 rows) can stay in
  // the XCD's 4-MB L2 while the A panels stream past it.
  // (divisors prepared on the host: prep_tile_map)
  auto map_tile = [&](int tid_lin, int& tn, int& tmg) {
    if (tid_lin < p.map_full_gsz) {
      const int ng = fdiv(tid_lin, p.fd_map_gsz), r = tid_lin - ng * p.map_gsz;
      tmg = fdiv(r, p.fd_map_gn);
      tn = ng * p.map_gn + (r - tmg * p.map_gn);
    } else {
      const int r = tid_lin - p.map_full_gsz;
      tmg = fdiv(r, p.fd_map_rn);
      tn = p.map_full * p.map_gn + (r - tmg * p.map_rn);
    }
  };
  int tile_n, tile_mg;
  map_tile(id, tile_n, tile_mg);
  int g = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroups; ++i)
    if (i < p.ngroups && tile_mg >= p.g_tile0[i]) g = i;
  const int g_row0 = MD_SEL_G(p.g_row0, g);
  const int g_arow0 = MD_SEL_G(p.g_arow0, g);
  const int m_base = g_row0 + (tile_mg - MD_SEL_G(p.g_tile0, g)) * BM;
  const int m_end = g_row0 + MD_SEL_G(p.g_rows, g);
  const int n0 = tile_n * BN;
  const char* Wg = (const char*)MD_SEL_G(p.W, g);
  const char* Ab = (const char*)p.A;
  long out_boff = 0;  // element offset of this batch in the output
  if (p.batch > 1) {
    const int by = blockIdx.y;
    const int bo = by / p.batch_inner, bi = by - bo * p.batch_inner;
    Ab += (bo * p.a_bs[0] + bi * p.a_bs[1]) * ESZ;
    Wg += (bo * p.w_bs[0] + bi * p.w_bs[1]) * ESZ;
    out_boff = bo * p.o_bs[0] + bi * p.o_bs[1];
  }

  // ---- per-lane global source pointers (swizzle applied on the source side) ----
  const int lrow = lane >> 3;  // row inside an 8-row group
  const int pc = lane & 7;     // physical 16-byte chunk inside the 128-byte row
  const char* srcA[A_ITERS];
  unsigned maskA[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int lc = pc ^ ((r >> 1) & 7);
    int m = m_base + r;
    m = m < m_end ? m : m_end - 1;
    const long am = (long)g_arow0 + (m - g_row0);  // physical A row
    if constexpr (AMODE == A_DENSE) {
      srcA[i] = Ab + (long)(int)am * (long)(int)(p.lda * ESZ) + lc * 16;  // 32 x 32 -> 64: one v_mad_i64_i32
      maskA[i] = 0;
    } else if constexpr (AMODE == A_INDEXED) {
      srcA[i] = Ab + (long)p.a_index[am] * (long)(int)(p.lda * ESZ) + lc * 16;
      maskA[i] = 0;
    } else {
      const int ow = p.cOW > 0 ? p.cOW : p.cW, oh = p.cOH > 0 ? p.cOH : p.cH;
      const int t2 = fdiv((int)am, p.fd_ow);
      const int ox = (int)am - t2 * ow;
      const long bimg = fdiv(t2, p.fd_oh);
      const int oy = t2 - (int)bimg * oh;
      const int x = ox * p.cstride, y = oy * p.cstride;  // centre tap in the input grid
      // tap mask = outer product of 3 row bits and 3 column bits; pixel index in 32 bits, ONE 64-bit multiply-add
      unsigned c3 = 0, mk = 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) c3 |= ((unsigned)(x + kx - 1) < (unsigned)p.cW ? 1u : 0u) << kx;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) mk |= ((unsigned)(y + ky - 1) < (unsigned)p.cH ? c3 : 0u) << (3 * ky);
      maskA[i] = mk;
      const int pix = ((int)bimg * p.cH + y) * p.cW + x;
      srcA[i] = Ab + (long)pix * (long)(p.cC * ESZ) + lc * 16;
    }
  }
  const long ldw = p.ldw > 0 ? p.ldw : (long)p.K;
  const char* srcW[W_ITERS];
#pragma unroll
  for (int i = 0; i < W_ITERS; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int lc = pc ^ ((r >> 1) & 7);
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    srcW[i] = Wg + (long)n * (long)(int)(ldw * ESZ) + lc * 16;
  }
  const char* zsrc = (const char*)p.zero_page + pc * 16;

  const int KT = p.K / KE;
  // conv bookkeeping: k-tile -> (tap, channel block); all wave-uniform
  const int cblocks = (AMODE == A_CONV3) ? (p.cCk > 0 ? p.cCk : p.cC) / KE : 1;

  auto issue = [&](int stage, int kt) {
    char* sbase = smem + stage * STAGE_BYTES;
    long a_delta;
    int tap = 0;
    if constexpr (AMODE == A_CONV3) {
      tap = fdiv(kt, p.fd_cblocks);
      int cb = kt - tap * cblocks;
      if (p.a_wrap > 0 && cb >= p.a_wrap) cb -= p.a_wrap;  // split-half operands: the third term re-reads A's hi plane
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;  // tap / 3 for tap < 9
      a_delta = ((long)(ky - 1) * p.cW + (kx - 1)) * p.cC * ESZ + (long)cb * 128;
    } else {
      const int kta = (p.a_wrap > 0 && kt >= p.a_wrap) ? kt - p.a_wrap : kt;
      a_delta = (long)kta * 128;
    }
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const char* s = srcA[i] + a_delta;
      if constexpr (AMODE == A_CONV3) {
        if (!((maskA[i] >> tap) & 1u)) s = zsrc;
      }
      glds16(s, sbase + (i * NW + wave) * 1024);
    }
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) {
      const int rgw = i * NW + wave;
      if (rgw < RG_W) glds16(srcW[i] + (long)kt * 128, sbase + BM * 128 + rgw * 1024);
    }
  };

  // 16x16 output tiles (v_mfma_f32_16x16x32 / 4 x v_mfma_f32_16x16x4_f32 / block-scaled 16x16x128): same FLOPs per LDS
  // byte as the 32x32 forms, and the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back item 7).
  // acc[a][b] = n16-tile a x m16-tile b of the wave tile: lane (r16 = lane & 15, q16 = lane >> 4) holds the 4 consecutive
  // columns n = 16a + 4 q16 .. + 3 of row m = 16b + r16 (the weight tile is the MFMA A operand).
  f32x4acc_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4acc_t){0.f, 0.f, 0.f, 0.f};

  const int q16 = lane >> 4, r16 = lane & 15;
  const int lane_off = r16 * 128 + ((((r16 >> 1) & 7) ^ q16) << 4);  // logical chunk 4 ks + q16, swizzled by (row >> 1) & 7

  const int KTg = KT / KSPLIT, kt0 = grp * KTg;  // this group's k-tiles (the launcher splits only when KSPLIT divides KT)
  issue(0, kt0);
  for (int i = 0; i < KTg; ++i) {
    const int kt = kt0 + i;
    const int cur = i & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i + 1 < KTg) issue(cur ^ 1, kt + 1);
    const char* As = smem + cur * STAGE_BYTES + wm * WTM * 128;
    const char* Ws = smem + cur * STAGE_BYTES + BM * 128 + wn * WTN * 128;
    if constexpr (std::is_same<T, fp8_t>::value) {  // block-scaled MFMA: both 16-byte halves of the k-tile per instruction
      i32x4_t af[2][TM], wf[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = lane_off ^ (ks << 6);
#pragma unroll
        for (int b = 0; b < TM; ++b) af[ks][b] = *(const i32x4_t*)(As + b * 2048 + off);
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[ks][a] = *(const i32x4_t*)(Ws + a * 2048 + off);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) mx_mma16(wf[0][a], wf[1][a], af[0][b], af[1][b], acc[a][b]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = lane_off ^ (ks << 6);
        i32x4_t af[TM], wf[TN];
#pragma unroll
        for (int b = 0; b < TM; ++b) af[b] = *(const i32x4_t*)(As + b * 2048 + off);
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = *(const i32x4_t*)(Ws + a * 2048 + off);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) Atom16<T>::mma(wf[a], af[b], acc[a][b]);
      }
    }
  }

  if constexpr (KSPLIT > 1) {
    // the groups' partial accumulators meet in the (now idle) rings: TN x TM vectors per lane, lane-contiguous; fixed order
    f32x4_t* const xch = (f32x4_t*)smem_all;
    __syncthreads();
    if (grp > 0) {
      f32x4_t* px = xch + ((grp - 1) * NW + wave) * (TN * TM * 64) + lane;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) px[(a * TM + b) * 64] = (f32x4_t){acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KSPLIT; ++g2) {
      const f32x4_t* px = xch + ((g2 - 1) * NW + wave) * (TN * TM * 64) + lane;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          const f32x4_t o = px[(a * TM + b) * 64];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b][j] += o[j];
        }
    }
  }

  if constexpr (BN == 64 && WGN == 2 && !std::is_same<T, fp8_t>::value) {
    if (p.epi == EPI_QKV && p.qkn_g[0] != nullptr && n0 < 2 * p.embed) {
      // ---- a q or k tile = one head: per-head LayerNorm(64) + 2-D RoPE on the accumulators (DA3's extended blocks; what
      //      qk_norm_rope_kernel does on the stored rows, here before the one rounding to T). A row's 64 columns live in the two
      //      waves wn = 0 | 1 of its wave row: the row statistics meet in the (idle) ring. ----
      const int isk = n0 >= p.embed ? 1 : 0;
      const float* bias = MD_SEL_G(p.bias, g);
      const int c0 = wn * 32 + 4 * q16;  // this lane's first column inside the head, for a = 0 (a = 1: + 16)
      f32x4_t y[TN][TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const f32x4_t bv = *(const f32x4_t*)(bias + n0 + c0 + 16 * a);
#pragma unroll
        for (int b = 0; b < TM; ++b) y[a][b] = (f32x4_t){acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]} + bv;
      }
      float* const red = (float*)smem_all;  // [2 passes][wn][BM]
      auto row_total = [&](float (&part)[TM], int pass) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          part[b] += __shfl_xor(part[b], 16);
          part[b] += __shfl_xor(part[b], 32);
          if (q16 == 0) red[(pass * 2 + wn) * BM + wm * WTM + b * 16 + r16] = part[b];
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < TM; ++b) part[b] = red[(pass * 2) * BM + wm * WTM + b * 16 + r16] + red[(pass * 2 + 1) * BM + wm * WTM + b * 16 + r16];
      };
      float mean[TM], rstd[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        mean[b] = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a) mean[b] += (y[a][b][0] + y[a][b][1]) + (y[a][b][2] + y[a][b][3]);
      }
      __syncthreads();  // every wave has left the main loop's last k-tile: the ring is free
      row_total(mean, 0);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        mean[b] *= (1.0f / 64.0f);
        rstd[b] = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          y[a][b] = y[a][b] - mean[b];
          rstd[b] += (y[a][b][0] * y[a][b][0] + y[a][b][1] * y[a][b][1]) + (y[a][b][2] * y[a][b][2] + y[a][b][3] * y[a][b][3]);
        }
      }
      row_total(rstd, 1);
      const f32x4_t g0 = *(const f32x4_t*)(p.qkn_g[isk] + c0), g1 = *(const f32x4_t*)(p.qkn_g[isk] + c0 + 16);
      const f32x4_t b0 = *(const f32x4_t*)(p.qkn_b[isk] + c0), b1 = *(const f32x4_t*)(p.qkn_b[isk] + c0 + 16);
      const float qs = isk ? 1.0f : p.qscale;
      const int two_d = 2 * p.embed;
      static_assert(TN == 2, "two n16-tiles per wave: the RoPE pair (j, j + 16) sits in one lane");
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const float rs = 1.0f / sqrtf(rstd[b] * (1.0f / 64.0f) + p.qkn_eps);
        const f32x4_t u0 = y[0][b] * rs * g0 + b0, u1 = y[1][b] * rs * g1 + b1;
        const int m = m_base + wm * WTM + b * 16 + r16;
        const int mc = m < m_end ? m : m_end - 1;
        const int t = mc - fdiv(mc, p.fd_seq_stride) * p.seq_stride;
        int py = 0, px = 0;
        if (t > 0 && t < p.rope_ntok) {
          if (p.rope_global) {
            py = px = 1;
          } else {
            const int pi = t - 1;
            py = fdiv(pi, p.fd_rope_pw);
            px = pi - py * p.rope_pw + 1;
            py += 1;
          }
        }
        const int pos = wn ? px : py;  // first half of the head rotates with the row position, the second with the column position
        const f32x4_t cs = *(const f32x4_t*)(p.rope_cos + pos * 16 + 4 * q16), sn = *(const f32x4_t*)(p.rope_sin + pos * 16 + 4 * q16);
        const f32x4_t o0 = (u0 * cs - u1 * sn) * qs, o1 = (u1 * cs + u0 * sn) * qs;
        if (m < m_end) {
          const int n = n0 + c0;
          if constexpr (is_split<TO>::value) {
            TO* row = (TO*)p.out + (long)m * (2L * two_d) + isk * two_d + (n - isk * p.embed);
            store4s<TO>(row, p.embed, o0);
            store4s<TO>(row + 16, p.embed, o1);
          } else {
            store4<TO>((TO*)p.out + (long)m * two_d + n, o0);
            store4<TO>((TO*)p.out + (long)m * two_d + n + 16, o1);
          }
        }
      }
      return;
    }
  }

  auto head_act = [&](float z) __attribute__((always_inline)) {
    return p.head_act == 1 ? expf(z) : (p.head_act == 2 ? z : (p.head_act == 3 ? expf(z) + 1.0f : fmaxf(z, 0.f)));
  };
  if (p.epi == EPI_HEAD) {
    // depth head tail (mod.rs:108-111): relu(conv1+b1) . w_out + b_out, relu. N == 32 == one tile: a lane holds 2 x 4 of a
    // row's 32 columns, the four q16 groups are summed with two cross-lane exchanges.
    if constexpr (BN == 32 && WGN == 1) {
      const float* bias = MD_SEL_G(p.bias, g);
      f32x4_t bv[TN], hw[TN];
#pragma unroll
      for (int a = 0; a < TN; ++a) bv[a] = *(const f32x4_t*)(bias + 16 * a + 4 * q16);
      auto act_of = [&](int kind, float z) __attribute__((always_inline)) {
        return kind == 1 ? expf(z) : (kind == 2 ? z : (kind == 3 ? expf(z) + 1.0f : fmaxf(z, 0.f)));
      };
      if (p.head_nch > 0) {
        // several output channels from ONE pass over the 3x3 convolution: relu(conv + b1) is formed once per element and
        // dotted with each channel's 32 weights (the same q16 reduction as the one-channel form)
        f32x4_t rl[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) rl[a][b][j] = fmaxf(acc[a][b][j] + bv[a][j], 0.f);
        for (int ch = 0; ch < p.head_nch; ++ch) {
#pragma unroll
          for (int a = 0; a < TN; ++a) hw[a] = *(const f32x4_t*)(p.head_wc[ch] + 16 * a + 4 * q16);
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            float part = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
              for (int j = 0; j < 4; ++j) part += rl[a][b][j] * hw[a][j];
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            const int m = m_base + wm * WTM + b * 16 + r16;
            if (q16 == 0 && m < m_end) {
              const int img = fdiv(m, p.fd_head_plane);
              p.head_out[ch][(long)img * p.head_bstride[ch] + (m - img * p.head_plane)] = act_of(p.head_acts[ch], part + p.head_bs[ch]);
            }
          }
        }
        return;
      }
#pragma unroll
      for (int a = 0; a < TN; ++a) hw[a] = *(const f32x4_t*)(p.head_w + 16 * a + 4 * q16);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        float part = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) part += fmaxf(acc[a][b][j] + bv[a][j], 0.f) * hw[a][j];
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const int m = m_base + wm * WTM + b * 16 + r16;
        if (q16 == 0 && m < m_end) ((float*)p.out)[m] = head_act(part + p.head_b);
      }
    }
    return;
  }
  if (p.epi == EPI_HEAD_UP2) {
    // Depth Pro's head behind the composed deconv -> conv1 (mod.rs:105-111; weights: compose_head_kernel): GEMM row m is
    // input pixel (img, y, x), the 32-column group 2*py + px is output pixel (2y+py, 2x+px). This wave's 64 columns are the
    // groups (py = wn, px = 0 | 1) = its n16-tiles {0,1} | {2,3}: one 8-byte store per row covers both px. The bias vector
    // is picked by the output pixel's position class (first / interior / last row x column), see compose_head_bias_kernel.
    if constexpr (AMODE == A_CONV3 && BN == 128 && WGN == 2) {
      const float* bias9 = MD_SEL_G(p.bias, g);
      f32x4_t hw4[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) hw4[e] = *(const f32x4_t*)(p.head_w + 16 * e + 4 * q16);
      const int H2 = 2 * p.cH, W2 = 2 * p.cW;
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m_base + wm * WTM + b * 16 + r16;
        const int mc = m < m_end ? m : m_end - 1;
        const int t2 = fdiv(mc, p.fd_ow);
        const int x = mc - t2 * p.cW;
        const int img = fdiv(t2, p.fd_oh);
        const int y = t2 - img * p.cH;
        const int Y = 2 * y + wn;
        const int ry = Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1);
        float z[2];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int X = 2 * x + px;
          const int rx = X == 0 ? 0 : (X == W2 - 1 ? 2 : 1);
          const float* bc = bias9 + (ry * 3 + rx) * 32 + 4 * q16;
          float part = 0.f;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x4_t bv = *(const f32x4_t*)(bc + 16 * e);
#pragma unroll
            for (int j = 0; j < 4; ++j) part += fmaxf(acc[2 * px + e][b][j] + bv[j], 0.f) * hw4[e][j];
          }
          part += __shfl_xor(part, 16);
          part += __shfl_xor(part, 32);
          z[px] = head_act(part + p.head_b);
        }
        if (q16 == 0 && m < m_end) {
          typedef __attribute__((ext_vector_type(2))) float f32x2_t;
          *(f32x2_t*)((float*)p.out + ((long)img * H2 + Y) * W2 + 2 * x) = (f32x2_t){z[0], z[1]};
        }
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = m_base + wm * WTM + b * 16 + r16;
      const int n = n0 + wn * WTN + a * 16 + 4 * q16;
      if (m < m_end && n < p.N) {
        const f32x4_t v = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        epilogue4<TO>(p, g, m, n, v, out_boff);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, BK = 128 bytes of K, 8 waves (2 x 4, wave tile 128 x 64) on v_mfma_f32_16x16x32 (bf16 / f16 / fp8) or
// 4 x v_mfma_f32_16x16x4_f32. LDS is a ring of FIVE 32-KB half-tile slots (160 KB, the whole CU): half-tile h = 2t is
// the A tile of k-tile t, h = 2t+1 its W tile, slot = h mod 5. While k-tile t is multiplied (2 slots busy) three more
// half-tiles (96 KB per CU) are in flight; the loads are never drained inside the loop -- counted `s_waitcnt vmcnt`
// and raw s_barrier. Staggered two-group schedule: waves 4..7 run one barrier behind waves 0..3 and every SIMD hosts
// one wave of each group, so one wave's 512-cycle MFMA cluster runs while its partner reads LDS / issues LDS-DMA.
// The epilogue goes through LDS (wave-private 64-row sub-tiles) so that every global instruction covers whole
// 128/256-byte row segments.
//
// EK selects the epilogue specialisation at compile time (a runtime switch around the unrolled row loops cost 40 VGPRs
// and scratch spills): 0 generic (epilogue4 per vector), 1 read-modify-write residual (EPI_RESID_LS), 2 2-byte store
// with optional residual inputs (EPI_STORE / q,k of EPI_QKV), 3 pixel shuffle (EPI_PIXSHUF), 4 = 2 with the GELU fused
// at compile time (the fc1 GEMM: its own kernel symbol, so profilers report it separately).
// LayerNorm folded into its neighbours (GemmParams::ln_*, round 6): 5 = 1 that also emits round_T(gamma_next . x_new) and the row
// statistics of x_new (per 256-column tile: mean and centred sum of squares, combined without cancellation), 6 = 2 / 7 = 4 that finish
// the accumulators with rstd_m * (acc - mu_m * c[n]) + d[n] before the store (q | k | V^T tiles of EPI_QKV; fc1's GELU).
// DIAG = true is the diagnostic build used by md_bench_gemm only (in-kernel stamps and timing-only ablation flags); the
// engine never launches it and the production instantiations contain none of that code.
// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), the total in every lane: rotations by 8, 4, 2, 1 add the same pairs
// in every lane, so all sixteen hold bit-identical totals
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
constexpr int kLnXchg = 8 * 64 * 272;  // LDS offset of the LayerNorm-fold exchange area: behind the eight wave-private staging regions (8 KB used)

template <typename T, int AMODE, int EK, bool DIAG>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
  // EK 8 / 9 / 10 / 11 = the lean 2-byte store kinds 2 / 4 / 6 / 7 in their DIRECT-store form (GemmParams::direct_store): its own
  // instantiation, so that the staged form's code and live ranges are not part of it (as a runtime branch of one kernel the pair sat
  // at 256 VGPRs and hipcc spilled LDS-DMA source addresses into the main loop's first k-tile)
  constexpr bool DS = EK >= 8;
  constexpr int EKB = EK == 8 ? 2 : (EK == 9 ? 4 : (EK == 10 ? 6 : (EK == 11 ? 7 : EK)));
  constexpr int BM = 256, BN = 256, NW = 8, WGN = 4;
  constexpr int WTM = 128, WTN = 64;
  constexpr int HALF_BYTES = 256 * 128;  // one half-tile slot
  constexpr int NSLOT = 5;
  typedef typename OutT<T>::type TO;  // element type of outputs / residuals
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KE = 128 / ESZ;
  constexpr int LPH = 4;  // glds wave-instructions per wave per half-tile (32 row groups / 8 waves)

  extern __shared__ __attribute__((aligned(16))) char smem[];

  unsigned long long* stamp = nullptr;
  if constexpr (DIAG) {
    stamp = (p.stamps && threadIdx.x == 0) ? p.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;
    if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int nwg = gridDim.x;
  int id;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // L2-aware raster: n-tiles are walked in groups of `gn` (host-chosen): the 32 blocks an XCD runs together cover
  // (32/gn) A panels x gn W panels (divisors prepared on the host: prep_tile_map)
  int tile_n, tile_mg;
  if (id < p.map_full_gsz) {
    const int ng = fdiv(id, p.fd_map_gsz), r = id - ng * p.map_gsz;
    tile_mg = fdiv(r, p.fd_map_gn);
    tile_n = ng * p.map_gn + (r - tile_mg * p.map_gn);
  } else {
    const int r = id - p.map_full_gsz;
    tile_mg = fdiv(r, p.fd_map_rn);
    tile_n = p.map_full * p.map_gn + (r - tile_mg * p.map_rn);
  }
  int g = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroups; ++i)
    if (i < p.ngroups && tile_mg >= p.g_tile0[i]) g = i;
  const int g_row0 = MD_SEL_G(p.g_row0, g);
  const int g_arow0 = MD_SEL_G(p.g_arow0, g);
  const int m_base = g_row0 + (tile_mg - MD_SEL_G(p.g_tile0, g)) * BM;
  const int m_end = g_row0 + MD_SEL_G(p.g_rows, g);
  const int n0 = tile_n * BN;
  const char* Wg = (const char*)MD_SEL_G(p.W, g);
  const char* Ab = (const char*)p.A;
  long out_boff = 0;
  if (p.batch > 1) {
    const int by = blockIdx.y;
    const int bo = by / p.batch_inner, bi = by - bo * p.batch_inner;
    Ab += (bo * p.a_bs[0] + bi * p.a_bs[1]) * ESZ;
    Wg += (bo * p.w_bs[0] + bi * p.w_bs[1]) * ESZ;
    out_boff = bo * p.o_bs[0] + bi * p.o_bs[1];
  }

  const int lrow = lane >> 3, pc = lane & 7;
  const char* srcA[LPH];
  unsigned maskA[LPH];
#pragma unroll
  for (int i = 0; i < LPH; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int lc = pc ^ ((r >> 1) & 7);
    int m = m_base + r;
    m = m < m_end ? m : m_end - 1;
    const long am = (long)g_arow0 + (m - g_row0);
    if constexpr (AMODE == A_DENSE) {
      srcA[i] = Ab + (long)(int)am * (long)(int)(p.lda * ESZ) + lc * 16;  // 32 x 32 -> 64: one v_mad_i64_i32
      maskA[i] = 0;
    } else if constexpr (AMODE == A_INDEXED) {
      srcA[i] = Ab + (long)p.a_index[am] * (long)(int)(p.lda * ESZ) + lc * 16;
      maskA[i] = 0;
    } else {
      const int ow = p.cOW > 0 ? p.cOW : p.cW, oh = p.cOH > 0 ? p.cOH : p.cH;
      const int t2 = fdiv((int)am, p.fd_ow);
      const int ox = (int)am - t2 * ow;
      const long bimg = fdiv(t2, p.fd_oh);
      const int oy = t2 - (int)bimg * oh;
      const int x = ox * p.cstride, y = oy * p.cstride;  // centre tap in the input grid
      // tap mask = outer product of 3 row bits and 3 column bits; pixel index in 32 bits, ONE 64-bit multiply-add
      unsigned c3 = 0, mk = 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) c3 |= ((unsigned)(x + kx - 1) < (unsigned)p.cW ? 1u : 0u) << kx;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) mk |= ((unsigned)(y + ky - 1) < (unsigned)p.cH ? c3 : 0u) << (3 * ky);
      maskA[i] = mk;
      const int pix = ((int)bimg * p.cH + y) * p.cW + x;
      srcA[i] = Ab + (long)pix * (long)(p.cC * ESZ) + lc * 16;
    }
  }
  const long ldw = p.ldw > 0 ? p.ldw : (long)p.K;
  // direct-store tiles (GemmParams::direct_store): LDS row r = 64 wn + 16 a + i of the W tile holds weight row
  // 64 wn + 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3): a lane's (a = 2h, 2h + 1) accumulators are then 8 consecutive columns
  // (launch_256 picks the direct form only for launches whose every non-V^T tile qualifies: no residual inputs, no second / fp32 / fp8
  // output, N % 256 == 0; the V^T tiles of a QKV launch keep the plain image and their transposed staging)
  const bool dstore = DS && !(p.epi == EPI_QKV && n0 >= 2 * p.embed);
  // W pieces as 32-bit byte offsets from the (wave-uniform) group base: a weight matrix is far below 4 GB, the LDS-DMA takes the base
  // from SGPRs (global_load_lds ... v_off, s[base]) and four VGPRs ride through the main loop instead of eight (round 6: the direct-store
  // kinds sat 4 registers over the budget and hipcc spilled source addresses into k-tile 0)
  unsigned offW[LPH];
#pragma unroll
  for (int i = 0; i < LPH; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int lc = pc ^ ((r >> 1) & 7);
    const int rp = (r & ~63) | (((r >> 5) & 1) << 5) | (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3);
    int n = n0 + (dstore ? rp : r);
    n = n < p.N ? n : p.N - 1;
    offW[i] = (unsigned)n * (unsigned)(ldw * ESZ) + (unsigned)(lc * 16);
  }
  const char* zsrc = (const char*)p.zero_page + pc * 16;
  const int KT = p.K / KE;
  const int cblocks = (AMODE == A_CONV3) ? (p.cCk > 0 ? p.cCk : p.cC) / KE : 1;

  // issue the A / W half-tile of k-tile kt into ring slot `slot` (both wave-uniform)
  bool freeze_k = false, no_loads = false;  // timing-only ablations of the diagnostic build
  if constexpr (DIAG) {
    freeze_k = (p.debug_flags & 2) != 0;  // every k-tile re-reads k-tile 0 (L2-resident)
    no_loads = (p.debug_flags & 1) != 0;
  }
  auto issue_W = [&](int kt, int slot) __attribute__((always_inline)) {
    char* sbase = smem + slot * HALF_BYTES;
    if (DIAG && freeze_k) kt = 0;
    const char* wk = Wg + (long)kt * 128;  // wave-uniform
#pragma unroll
    for (int i = 0; i < LPH; ++i) glds16(wk + offW[i], sbase + (i * NW + wave) * 1024);
  };
  auto issue_A = [&](int kt, int slot) __attribute__((always_inline)) {
    char* sbase = smem + slot * HALF_BYTES;
    if (DIAG && freeze_k) kt = 0;
    long a_delta;
    int tap = 0;
    if constexpr (AMODE == A_CONV3) {
      tap = fdiv(kt, p.fd_cblocks);
      int cb = kt - tap * cblocks;
      if (p.a_wrap > 0 && cb >= p.a_wrap) cb -= p.a_wrap;  // split-half operands: the third term re-reads A's hi plane
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;  // tap / 3 for tap < 9
      a_delta = ((long)(ky - 1) * p.cW + (kx - 1)) * p.cC * ESZ + (long)cb * 128;
    } else {
      const int kta = (p.a_wrap > 0 && kt >= p.a_wrap) ? kt - p.a_wrap : kt;
      a_delta = (long)kta * 128;
    }
#pragma unroll
    for (int i = 0; i < LPH; ++i) {
      const char* s = srcA[i] + a_delta;
      if constexpr (AMODE == A_CONV3) {
        if (!((maskA[i] >> tap) & 1u)) s = zsrc;
      }
      glds16(s, sbase + (i * NW + wave) * 1024);
    }
  };

  // acc16[n16-tile][m16-tile], lane (r = lane & 15, q = lane >> 4) holds 4 consecutive columns n of one row m
  f32x4acc_t acc16[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc16[a][b] = (f32x4acc_t){0.f, 0.f, 0.f, 0.f};
  const int q16 = lane >> 4;
  const int lane_off16 = (lane & 15) * 128 + (((((lane & 15) >> 1) & 7) ^ q16) << 4);

  constexpr bool ln_cons = EKB == 6 || EKB == 7, ln_emit = EKB == 5, gelu_k = EKB == 4 || EKB == 7;
  // half-tile order: A0 W0 A1 W1 A2 | W2 A3 | W3 A4 | ...   (slot = order index mod 5)
  if (DIAG && stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
  int issued = 2, slot_i = 2;
  issue_A(0, 0);
  issue_W(0, 1);
  // SHORT prologue: only k-tile 0 is requested up front; A1, W1, A2 are issued from the load segments of k-tile 0 (every
  // ring slot is free then). In-kernel stamps showed the five-half-tile prologue costing 1.9 us of issue + 1.8 us until
  // k-tile 0 had landed behind the other 96 KB of the queue. Short-K tiles (KT <= 4: the deconvolutions, K = 128 / 256)
  // request everything up front instead: with two k-tiles the in-loop fill exposed a second full memory latency.
  const bool pro_full = KT <= 4;
  if (pro_full) {
    if (KT > 1) {
      issue_A(1, 2);
      issue_W(1, 3);
      issued = 4;
      slot_i = 4;
    }
    if (KT > 2) {
      issue_A(2, 4);
      issued = 5;
      slot_i = 0;
    }
  }
  int slot_c = 0;  // slot of the current tile's A half

  // counted wait for the two half-tiles of k-tile `t`: all but the `younger` most recent half-tiles
  auto wait_tile = [&](int t) __attribute__((always_inline)) {
    const int younger = issued - (2 * t + 2);
    if (younger >= 3) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (younger == 2) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (younger == 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  auto issue_next_W = [&](int t) __attribute__((always_inline)) {
    if (t >= 1 && t + 1 < KT && !(DIAG && no_loads)) {
      issue_W(t + 1, slot_i);
      ++issued;
      slot_i = slot_i == NSLOT - 1 ? 0 : slot_i + 1;
    }
  };
  auto issue_next_A = [&](int t) __attribute__((always_inline)) {
    if (t >= 1 && t + 2 < KT && !(DIAG && no_loads)) {
      issue_A(t + 2, slot_i);
      ++issued;
      slot_i = slot_i == NSLOT - 1 ? 0 : slot_i + 1;
    }
  };

  // LayerNorm-fold consumer (EK 6 / 7): the tile's 256 (rstd, -mu rstd) pairs (2 KB of p.ln_stats) arrive by LDS-DMA -- no VGPR rides
  // through the main loop, no compiler-placed wait: waves 0 and 1 request them in the load segment of the LAST k-tile into the exchange
  // area at kLnXchg (inside ring slot 4, behind every wave's staging region), when that slot is idle then (its last reader was k-tile
  // KT - 2 or earlier: true for KT = 16 and 32); otherwise behind the loop's last barrier. Three earlier forms -- the partials combined
  // behind the main loop; in the prologue with two VGPRs through the loop (hipcc spilled an address register whose reload in k-tile 0
  // waits vmcnt(0): the LDS-DMA pipeline drained once per tile); eight 8-byte loads per lane in the epilogue -- each cost 2.3 - 3.8 us
  // of a 32-us fc1 tile: the epilogue's first microsecond is latency-bound.
  const bool ln_in_loop = ln_cons && KT >= 2 && (2 * KT - 2) % NSLOT != 4 && (2 * KT - 1) % NSLOT != 4;
  // ln_raw (round 6, last form): p.ln_stats holds the producer's PARTIALS ([rows][4][2]: mean and centred sum of squares per 256-column
  // tile); every wave requests the 32 rows x 32 bytes of its share (8 KB per tile, behind the 2 KB of pairs in the exchange area) and the
  // epilogue's threads 0 .. 255 combine them into the pairs -- the ln_finish launch between the two GEMMs (47 per ViT pass, 6.5 us each
  // whatever the batch: 1.3 % of a B = 1 frame) is gone.
  const bool ln_raw = ln_cons && p.ln_raw != 0;
  auto issue_ln_ab = [&]() __attribute__((always_inline)) {
    if (ln_raw) {
      int l2 = lane;
      asm volatile("" : "+v"(l2));  // keeps the address arithmetic (and its two VGPRs) out of the main loop's live ranges
      int r2 = m_base + wave * 32 + (l2 >> 1);
      r2 = r2 < m_end - 1 ? r2 : m_end - 1;  // rows past the group are never used; keep the request inside the array
      glds16((const char*)p.ln_stats + (long)r2 * 32 + (l2 & 1) * 16, smem + kLnXchg + 2048 + wave * 1024);
    } else if (wave < 2) {
      int l2 = lane;
      asm volatile("" : "+v"(l2));
      int r2 = m_base + 2 * (wave * 64 + l2);
      r2 = r2 < m_end - 2 ? r2 : m_end - 2;
      glds16((const char*)p.ln_stats + (long)r2 * 8, smem + kLnXchg + wave * 1024);
    }
  };
  // ---- the staggered two-group schedule on 16x16x32 MFMAs, with 512-cycle MFMA clusters ----
  // Two phases per k-tile (one per 32-deep k-step): R = 4 W + 8 A fragment reads (+ the LDS-DMA issue of the next
  // half-tiles), M = 32 MFMAs = 512 cycles, so the ~100-cycle s_barrier round trip is paid 4 times per k-tile. Every R
  // ends with lgkmcnt(0) BEFORE its barrier (R has 512 cycles of cover), so the slots of tile t are free one interval
  // after its last R and the loads that reuse them go out in R(0) of tile t+1. Both groups stay at priority 0
  // (s_setprio 1 around the cluster cost 0.65 % of the step).
  const bool g1 = wm == 1;
  if (DIAG && stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
  wait_tile(0);
  __builtin_amdgcn_s_barrier();
  if (DIAG && stamp) {
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    stamp[8] = __builtin_readcyclecounter();
  }
  if (g1) __builtin_amdgcn_s_barrier();
  if constexpr (std::is_same<T, fp8_t>::value) {
    // e4m3 operands: ONE phase per 128-deep k-tile -- R = 8 W + 16 A fragment reads (both 64-deep halves; a block-scaled
    // MFMA consumes a half of each) + both LDS-DMA issues, M = 32 x v_mfma_scale_f32_16x16x128_f8f6f4 = 1024 cycles.
    for (int t = 0; t < KT; ++t) {
      const int slot_w = slot_c == NSLOT - 1 ? 0 : slot_c + 1;
      const char* As = smem + slot_c * HALF_BYTES + wm * WTM * 128;
      const char* Ws = smem + slot_w * HALF_BYTES + wn * WTN * 128;
      slot_c = slot_w == NSLOT - 1 ? 0 : slot_w + 1;
      i32x4_t wf[2][4], af[2][8];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = lane_off16 ^ (ks << 6);
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[ks][a] = *(const i32x4_t*)(Ws + a * 2048 + off);
#pragma unroll
        for (int b = 0; b < 8; ++b) af[ks][b] = *(const i32x4_t*)(As + b * 2048 + off);
      }
      if (t == 0) {  // rest of the pipeline fill (see the prologue)
        if (!(DIAG && no_loads) && !pro_full) {
          if (KT > 1) { issue_A(1, 2); issue_W(1, 3); issued = 4; slot_i = 4; }
          if (KT > 2) { issue_A(2, 4); issued = 5; slot_i = 0; }
        }
      } else {
        issue_next_W(t);
        issue_next_A(t);
      }
      if (g1 && t + 1 < KT) wait_tile(t + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) mx_mma16(wf[0][a], wf[1][a], af[0][b], af[1][b], acc16[a][b]);
      if (!g1 && t + 1 < KT) wait_tile(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  for (int t = 0; t < KT; ++t) {
    const int slot_w = slot_c == NSLOT - 1 ? 0 : slot_c + 1;
    const char* As = smem + slot_c * HALF_BYTES + wm * WTM * 128;
    const char* Ws = smem + slot_w * HALF_BYTES + wn * WTN * 128;
    slot_c = slot_w == NSLOT - 1 ? 0 : slot_w + 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = lane_off16 ^ (ks << 6);
      i32x4_t wf[4], af[8];
#pragma unroll
      for (int a = 0; a < 4; ++a) wf[a] = *(const i32x4_t*)(Ws + a * 2048 + off);
#pragma unroll
      for (int b = 0; b < 8; ++b) af[b] = *(const i32x4_t*)(As + b * 2048 + off);
      if (t == 0) {  // rest of the pipeline fill, in half-tile order A1 W1 A2 (slots 2, 3, 4)
        if (!(DIAG && no_loads) && !pro_full) {
          if (ks == 0 && KT > 1) { issue_A(1, 2); issue_W(1, 3); issued = 4; slot_i = 4; }
          if (ks == 1 && KT > 2) { issue_A(2, 4); issued = 5; slot_i = 0; }
        }
      } else {
        if (ks == 0) issue_next_W(t);
        if (ks == 1) issue_next_A(t);
      }
      if constexpr (ln_cons) {
        if (ks == 0 && t == KT - 1 && ln_in_loop) issue_ln_ab();
      }
      if (g1 && ks == 1 && t + 1 < KT) wait_tile(t + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) Atom16<T>::mma(wf[a], af[b], acc16[a][b]);
      if (!g1 && ks == 1 && t + 1 < KT) wait_tile(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
  if (!g1) __builtin_amdgcn_s_barrier();

  // ---------------- epilogue ----------------
  if (DIAG && stamp) {
    stamp[4] = __builtin_amdgcn_s_memrealtime();
    stamp[9] = __builtin_readcyclecounter();
  }
  auto stamp_end = [&]() __attribute__((always_inline)) {
    if (DIAG && stamp) {
      stamp[6] = __builtin_amdgcn_s_memrealtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store drain of this wave (s_endpgm waits for it too)
      stamp[7] = __builtin_amdgcn_s_memrealtime();
    }
  };
  float* const lnx = (float*)(smem + kLnXchg);
  // LayerNorm-fold consumer (EK 6 / 7): p.ln_stats = [rows][2] fp32 (rstd, -mu rstd) per row, finished from the producer's partials by
  // ln_finish_kernel between the two GEMMs. (Two earlier forms: the partials combined here behind the main loop -- dependent loads, a
  // division and a square root in front of the epilogue: +2.4 us per 32-us fc1 tile; combined in the prologue and carried through the
  // main loop in two VGPRs -- the kernel sits at the 256-register edge, hipcc spilled an address register and its reload in k-tile 0
  // waits vmcnt(0), draining the LDS-DMA pipeline once per tile: the same +2.3 us.)
  if constexpr (ln_cons) {
    if (!ln_in_loop) {  // slot 4 was the last k-tile's: request the pairs once every wave has left the ring
      __builtin_amdgcn_s_barrier();
      issue_ln_ab();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requesting waves' pairs have landed; each path's first barrier publishes them
    if (ln_raw) {  // partials -> pairs (Chan's combination: every term of the variance is non-negative), one row per thread
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (tid < BM) {
        const f32x4_t p0 = *(const f32x4_t*)(lnx + 512 + tid * 8), p1 = *(const f32x4_t*)(lnx + 512 + tid * 8 + 4);
        const float mu = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
        const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
        const float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + 256.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        const float rstd = 1.0f / sqrtf(m2 * p.ln_inv_n + p.ln_eps);
        *(f32x2_t*)(lnx + 2 * tid) = (f32x2_t){rstd, -mu * rstd};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const bool direct = (p.epi == EPI_QKV && n0 >= 2 * p.embed);  // V^T wants lanes along tokens
  if (direct) {
    if constexpr (sizeof(TO) == 2) {
      // V^T[seq][head][d][token] tiles: staged TRANSPOSED through the wave-private LDS area (64 n-rows x 64 tokens
      // per half) so that a lane stores 4 consecutive tokens of one (head, d) row -- 8-byte stores, 4 full 128-byte
      // row segments per instruction -- instead of 2-byte scatter stores (4x the store instructions). Needs whole
      // 64-column head slices (embed % 64 == 0, always true: head_dim = 64) and 4-token groups inside one sequence
      // (seq_stride % 4 == 0).
      if ((p.seq_stride & 3) == 0 && n0 + BN <= p.N) {
        constexpr int SRT = 272;  // 64 tokens x 4 B + 16 B pad: the 4 q16 groups of a write land 16 banks apart
        char* stt = smem + wave * (64 * SRT);
        __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
        const int nl0 = lane >> 4;                       // n row of iteration 0
        const int c0 = n0 - 2 * p.embed + wn * WTN;      // first V column of this wave: one whole head
        const int hd = c0 >> 6;
        const float* bp_ = MD_SEL_G(p.bias, g) + n0 + wn * WTN;
        const float* wsp_ = MD_SEL_G(p.wscale, g);
        float bia[16], wsc[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          bia[it] = bp_[it * 4 + nl0];
          if constexpr (ln_cons) wsc[it] = MD_SEL_G(p.ln_c, g)[n0 + wn * WTN + it * 4 + nl0];  // c[n]; bia = d[n]
          else wsc[it] = wsp_ ? wsp_[n0 + wn * WTN + it * 4 + nl0] * p.ascale : 1.f;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const f32x4acc_t c = acc16[a][half * 4 + bb];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                *(float*)(stt + (a * 16 + 4 * q16 + j) * SRT + (bb * 16 + (lane & 15)) * 4) = c[j];
            }
          asm volatile("" ::: "memory");
          const int m = m_base + wm * WTM + half * 64 + (lane & 15) * 4;  // first of this lane's 4 tokens
          f32x4_t lnA4 = {1.f, 1.f, 1.f, 1.f}, lnB4 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (ln_cons) {  // (rstd, -mu rstd) of the lane's 4 tokens (rows m .. m + 3: all inside or all outside the group)
            const float* ab = lnx + 2 * (wm * WTM + half * 64 + (lane & 15) * 4);
            const f32x4_t t0 = *(const f32x4_t*)ab, t1 = *(const f32x4_t*)(ab + 4);
            lnA4 = (f32x4_t){t0[0], t0[2], t1[0], t1[2]};
            lnB4 = (f32x4_t){t0[1], t0[3], t1[1], t1[3]};
          }
          const int seq = fdiv(m, p.fd_seq_stride);
          const int tok = m - seq * p.seq_stride;
          TO* vrow = (TO*)p.vT + (((long)seq * p.heads + hd) * 64) * p.kpad + tok;
          const bool ok = m < m_end;  // groups of 4 rows are all inside or all outside (g_rows % 4 == 0 for token buffers)
#pragma unroll
          for (int it = 0; it < 16; ++it) {
            const int nl = it * 4 + nl0;
            f32x4_t v = *(const f32x4_t*)(stt + nl * SRT + (lane & 15) * 16);
            if constexpr (ln_cons) v = fma4(v, lnA4, fma4(lnB4, (f32x4_t){wsc[it], wsc[it], wsc[it], wsc[it]}, (f32x4_t){bia[it], bia[it], bia[it], bia[it]}));
            else v = v * wsc[it] + bia[it];
            if (ok) store4p<TO>(vrow + (long)nl * p.kpad, p.v_plane, v);  // split-half: the lo plane v_plane elements behind
          }
          asm volatile("" ::: "memory");
        }
        return;
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int m = m_base + wm * WTM + b * 16 + (lane & 15);
        const int n = n0 + wn * WTN + a * 16 + 4 * q16;
        if (m < m_end && n < p.N) {
          f32x4_t v = {acc16[a][b][0], acc16[a][b][1], acc16[a][b][2], acc16[a][b][3]};
          epilogue4<TO>(p, g, m, n, v, out_boff);
        }
      }
    return;
  }
  // staged: wave-private region of 64 rows x 272 bytes (68 floats).  Only the first barrier is a block
  // barrier (every wave is done reading the ring); the staging itself is wave-private and DS operations of
  // one wave execute in order.
  constexpr int SROW = 272;
  char* st = smem + wave * (64 * SROW);
  __builtin_amdgcn_s_barrier();
  if (DIAG && stamp) stamp[5] = __builtin_amdgcn_s_memrealtime();
  const int col = (lane & 15) * 4;
  const int n = n0 + wn * WTN + col;
  const bool nvalid = n < p.N;
  constexpr bool rmw = EKB == 1 || EKB == 5;
  constexpr bool fast_store = (EKB == 2 || EKB == 4 || EKB == 6 || EKB == 7) && sizeof(TO) == 2;  // EK 4 / 7: GELU fused at compile time (fc1)
  constexpr bool pixshuf = EKB == 3;
  const float* biasp = MD_SEL_G(p.bias, g);
  const int r16 = lane & 15;
  const int c8 = (lane & 7) * 8, rsub = lane >> 3;
  const int n8 = n0 + wn * WTN + c8;
  const bool nv8 = n8 < p.N;
  const bool interior = (m_base + BM <= m_end) && (n0 + BN <= p.N);  // wave-uniform: no per-row predicates needed
  // bf16 / f16 staging of a 64-row half: 128-byte rows, 16-byte chunks XOR-swizzled by (row & 7): conflict-free for the
  // 8-byte writes from the accumulator layout and for the 16-byte reads (a lane then owns 8 consecutive columns)
  // (split-half outputs: the hi plane's image at st, the lo plane's 8 KB behind it -- 16 of the wave's 17 KB)
  auto stage_half_2b = [&](int half, auto&& xform) __attribute__((always_inline)) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4acc_t c = acc16[a][half * 4 + bb];
        const f32x4_t v = xform(a, half * 4 + bb, (f32x4_t){c[0], c[1], c[2], c[3]});
        const int row = bb * 16 + r16;
        const int chunk = (a * 2 + (q16 >> 1)) ^ (row & 7);
        if constexpr (is_split<TO>::value) {
          i32x2_t hi, lo;
          split4<TO>(v, hi, lo);
          *(i32x2_t*)(st + row * 128 + chunk * 16 + (q16 & 1) * 8) = hi;
          *(i32x2_t*)(st + 8192 + row * 128 + chunk * 16 + (q16 & 1) * 8) = lo;
        } else {
          *(i32x2_t*)(st + row * 128 + chunk * 16 + (q16 & 1) * 8) = pack4<TO>(v);
        }
      }
  };
  auto stage_half_f32 = [&](int half) __attribute__((always_inline)) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4acc_t c = acc16[a][half * 4 + bb];
        *(f32x4_t*)(st + (bb * 16 + r16) * SROW + (a * 16 + 4 * q16) * 4) = (f32x4_t){c[0], c[1], c[2], c[3]};
      }
  };
  if constexpr (pixshuf && sizeof(TO) == 2) {
    // ---- pixel shuffle, fast form (launch_gemm sets ps_fast) ----
    // out row of input pixel m = y'*psW + x (y' = b*psH + y) and tap (dy, dx): (f*y' + dy)*(f*psW) + f*x + dx
    //   = f*m + f*(f-1)*psW*y' + (dy*f*psW + dx): ONE multiply-high division per row instead of two divisions and a
    // chain of 64-bit products. Bias is added in the accumulator layout and a lane stores 8 channels = 16 bytes of one
    // output pixel.
    if (p.ps_fast) {
      const int f = p.ps_f;
      const int tap8 = fdiv(nv8 ? n8 : 0, p.fd_psC);
      const int co8 = (nv8 ? n8 : 0) - tap8 * p.psC;
      const int dy8 = f == 4 ? tap8 >> 2 : tap8 >> 1, dx8 = tap8 - dy8 * f;
      const unsigned ldo_u = (unsigned)p.ldo;
      const unsigned c_lane = (unsigned)(dy8 * f * p.psW + dx8) * ldo_u + (unsigned)(p.ps_coff + co8);
      const unsigned k1 = (unsigned)f * ldo_u, k2 = (unsigned)(f * (f - 1) * p.psW) * ldo_u;
      f32x4_t bq[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int na = n0 + wn * WTN + a * 16 + 4 * q16;
        const int ca = na - fdiv(na, p.fd_psC) * p.psC;
        bq[a] = (biasp && na < p.N) ? *(const f32x4_t*)(biasp + ca) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
      char* ob = (char*)p.out;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        asm volatile("" ::: "memory");
        stage_half_2b(half, [&](int a, int, f32x4_t v) { return v + bq[a]; });
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + rsub;
          const int m = m_base + wm * WTM + half * 64 + row;
          const i32x4_t raw = *(const i32x4_t*)(st + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
          if (interior || (m < m_end && nv8)) {
            const unsigned yq = (unsigned)fdiv(m, p.fd_psW);
            const unsigned eo = (unsigned)m * k1 + yq * k2 + c_lane;
            *(i32x4_t*)(ob + (size_t)eo * 2u) = raw;
            if constexpr (is_split<TO>::value) {  // the lo plane's image sits 8 KB behind the hi plane's; its columns o_plane further right
              const i32x4_t raw_lo = *(const i32x4_t*)(st + 8192 + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
              *(i32x4_t*)(ob + ((size_t)eo + (size_t)p.o_plane) * 2u) = raw_lo;
            }
          }
        }
        asm volatile("" ::: "memory");
      }
      stamp_end();
      return;
    }
  }
  if constexpr (fast_store) {
    // ---- 2-byte store epilogue (EPI_STORE / q,k tiles of EPI_QKV; launcher guarantees N, ldo, ldr % 8 == 0) ----
    // A lane owns 8 consecutive columns of a row: per 64-row half 8 iterations of {LDS read, math, ONE 16-byte store};
    // runtime options are folded into wave-uniform flags once, and interior tiles run without per-row predicates.
    constexpr int PLN = kPlanes<TO>;  // split-half rows: [hi | lo] per section
    const long ldo8 = p.epi == EPI_QKV ? 2L * p.embed * PLN : p.ldo;
    // element offset from a value's hi to its lo plane, and the tile's first column inside the physical row: q | k rows of the
    // split-half form are [q_hi | q_lo | k_hi | k_lo], each `embed` wide (the launcher keeps a tile inside one section)
    const long lo_off = PLN == 1 ? 0 : (p.epi == EPI_QKV ? (long)p.embed : p.o_plane);
    const long n0p = (PLN == 2 && p.epi == EPI_QKV) ? (n0 >= p.embed ? n0 + p.embed : n0) : n0;
    const bool f32o = p.out_f32 != 0, relu = p.act == ACT_RELU, has_o2 = p.out2 != nullptr;
    const bool r1 = p.res1 != nullptr, r2 = p.res2 != nullptr, any_res = r1 || r2;
    const float* wsp = MD_SEL_G(p.wscale, g);  // fp8 operands: acc * (ascale * wscale[n]) before the bias
    const bool fp8o = p.out_fp8 != 0;
    const long tb = (long)m_base * ldo8 + n0p + out_boff;
    char* ob = (char*)p.out + tb * (fp8o ? 1 : (f32o ? 4 : 2));
    char* o2b = (char*)p.out2 + tb * 2;
    const unsigned lc8 = (unsigned)(wn * WTN + c8);
    // No residual inputs and a plain 2-byte output (fc1, the q/k tiles of qkv, most convolutions): bias, scale and
    // activation are applied in the ACCUMULATOR layout and the tile is staged in the output type -- half the LDS bytes
    // of the fp32 staging (LDS bandwidth was half of that epilogue's 4.2 us).
    if (!any_res && !f32o && !fp8o && !has_o2) {
      f32x4_t bq[4], wq[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        // the 4 columns of the lane's accumulators of n-block a: consecutive from here (direct-store tiles: the permuted image)
        const int na = dstore ? n0 + wn * WTN + (a >> 1) * 32 + q16 * 8 + (a & 1) * 4 : n0 + wn * WTN + a * 16 + 4 * q16;
        bq[a] = (biasp && na < p.N) ? *(const f32x4_t*)(biasp + na) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        wq[a] = (wsp && na < p.N) ? *(const f32x4_t*)(wsp + na) * p.ascale : (f32x4_t){1.f, 1.f, 1.f, 1.f};
        if constexpr (ln_cons) wq[a] = na < p.N ? *(const f32x4_t*)(MD_SEL_G(p.ln_c, g) + na) : (f32x4_t){0.f, 0.f, 0.f, 0.f};  // c[n]; bq = d[n]
        if (p.epi == EPI_QKV && na < p.embed) {  // q columns: (acc * w + b) * qscale, folded into the two vectors
          bq[a] *= p.qscale;
          wq[a] *= p.qscale;
        }
      }
      // LayerNorm fold: out = acc * rstd_m + (c[n] * (-mu_m rstd_m) + d[n]), (x qscale on q tiles: a tile lies inside one section);
      // the lane's eight rows are r16 of the m-blocks 0 .. 7
      float lnA[8], lnB[8];
      if constexpr (ln_cons) {
        const float qs_t = (p.epi == EPI_QKV && n0 < p.embed) ? p.qscale : 1.f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const f32x2_t t = *(const f32x2_t*)(lnx + 2 * (wm * WTM + b * 16 + r16));
          lnA[b] = t[0] * qs_t;
          lnB[b] = t[1];
        }
      }
      auto finish = [&](int a, int b, f32x4_t v) __attribute__((always_inline)) {
        if constexpr (ln_cons) {
          const f32x4_t Bv = {lnB[b], lnB[b], lnB[b], lnB[b]}, Av = {lnA[b], lnA[b], lnA[b], lnA[b]};
          v = fma4(v, Av, fma4(wq[a], Bv, bq[a]));
        } else {
          v = fma4(v, wq[a], bq[a]);
        }
        if constexpr (gelu_k) {
          v = gelu4<TO>(v);
        } else if (relu) {
          v = relu4(v);
        }
        return v;
      };
      if constexpr (DS) {
        // straight from the accumulator layout: lane (r16, q16) owns row b * 16 + r16 of m-block b and, for h = 0, 1, the 8 columns
        // 64 wn + 32 h + 8 q16 .. + 7 (n-blocks 2h and 2h + 1 of the permuted image)
        // (lane indices made opaque here: hipcc otherwise hoists the sixteen store offsets above the main loop and carries them through it)
        int r16e = r16, q16e = q16;
        asm volatile("" : "+v"(r16e), "+v"(q16e));
        const unsigned lcd = (unsigned)(wn * WTN + q16e * 8);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int lrow_t = wm * WTM + b * 16 + r16e;
          const bool ok = interior || m_base + lrow_t < m_end;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4acc_t c0 = acc16[2 * h][b], c1 = acc16[2 * h + 1][b];
            const f32x4_t v0 = finish(2 * h, b, (f32x4_t){c0[0], c0[1], c0[2], c0[3]});
            const f32x4_t v1 = finish(2 * h + 1, b, (f32x4_t){c1[0], c1[1], c1[2], c1[3]});
            const unsigned eo = ((unsigned)lrow_t * (unsigned)ldo8 + lcd + (unsigned)(h * 32)) * 2u;
            if constexpr (PLN == 2) {
              i32x4_t ph, pl;
              split8<TO>(v0, v1, ph, pl);
              if (ok) {
                *(i32x4_t*)(ob + eo) = ph;
                *(i32x4_t*)(ob + eo + (unsigned)lo_off * 2u) = pl;
              }
            } else {
              const i32x4_t raw = pack8<TO>(v0, v1);
              if (ok) *(i32x4_t*)(ob + eo) = raw;
            }
          }
          // one m-block at a time: left free, the scheduler interleaves all sixteen (b, h) groups, the epilogue's pressure passes 256
          // registers and the allocator answers by spilling LDS-DMA source addresses whose reloads land in the main loop's first k-tile
          __builtin_amdgcn_sched_barrier(0);
        }
        stamp_end();
        return;
      } else {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        asm volatile("" ::: "memory");
        if (!(DIAG && (p.debug_flags & 8)))  // timing-only ablation: skip the staging writes
          stage_half_2b(half, finish);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + rsub;
          const int lrow_t = wm * WTM + half * 64 + row;
          const i32x4_t raw = *(const i32x4_t*)(st + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
          if (DIAG && (p.debug_flags & 4)) {  // timing-only ablation: no global stores (keep the value alive)
            if (raw[0] == 0x12345678) *(i32x4_t*)(ob + ((unsigned)lrow_t * (unsigned)ldo8 + lc8) * 2u) = raw;
          } else if (interior || (m_base + lrow_t < m_end && nv8)) {
            *(i32x4_t*)(ob + ((unsigned)lrow_t * (unsigned)ldo8 + lc8) * 2u) = raw;
            if constexpr (PLN == 2) {
              const i32x4_t raw_lo = *(const i32x4_t*)(st + 8192 + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
              *(i32x4_t*)(ob + ((unsigned)lrow_t * (unsigned)ldo8 + lc8 + (unsigned)lo_off) * 2u) = raw_lo;
            }
          }
        }
        asm volatile("" ::: "memory");
      }
      stamp_end();
      return;
      }
    }
    if constexpr (DS) return;  // (the direct kinds are launched for lean launches only)
    // residual inputs / fp32, fp8 or second output: fp32 staging, the raw residual vectors of a half prefetched before
    // its staging pass (the dependent load -> store chain, not bandwidth, set the cost of the residual-conv epilogue)
    // Split-half outputs (round 6; they took the generic per-vector epilogue before: dec_conv3x3 2.24x its bf16 time): the same
    // fp32 staging, both planes of the residual inputs through a rolling window of four row groups (a value is its hi + lo), both
    // planes of the result (and of the relu'd second output) by 16-byte stores.
    if constexpr (is_split<TO>::value) {
      f32x4_t bl = {0.f, 0.f, 0.f, 0.f}, bh = bl;
      if (biasp && nv8) {
        bl = *(const f32x4_t*)(biasp + n8);
        bh = *(const f32x4_t*)(biasp + n8 + 4);
      }
      const long trb = (long)m_base * p.ldr + n0;
      const char* q1b = (const char*)p.res1 + trb * 2;
      const char* q2b = (const char*)p.res2 + trb * 2;
      const unsigned rlo = (unsigned)p.r_plane * 2u, olo = (unsigned)lo_off * 2u;
      i32x4_t w1h[4], w1l[4], w2h[4], w2l[4];
      auto issue_res = [&](int half, int it) __attribute__((always_inline)) {
        if (!any_res) return;
        const int lrow_t = wm * WTM + half * 64 + it * 8 + rsub;
        const bool ok = interior || (m_base + lrow_t < m_end && nv8);
        const unsigned ro = ((unsigned)lrow_t * (unsigned)p.ldr + lc8) * 2u;
        const i32x4_t z = {0, 0, 0, 0};
        w1h[it & 3] = (r1 && ok) ? *(const i32x4_t*)(q1b + ro) : z;
        w1l[it & 3] = (r1 && ok) ? *(const i32x4_t*)(q1b + ro + rlo) : z;
        w2h[it & 3] = (r2 && ok) ? *(const i32x4_t*)(q2b + ro) : z;
        w2l[it & 3] = (r2 && ok) ? *(const i32x4_t*)(q2b + ro + rlo) : z;
      };
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        asm volatile("" ::: "memory");
        stage_half_f32(half);
        asm volatile("" ::: "memory");
        if (half == 0) {
#pragma unroll
          for (int it = 0; it < 4; ++it) issue_res(0, it);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + rsub;
          const int lrow_t = wm * WTM + half * 64 + row;
          f32x4_t lo = *(const f32x4_t*)(st + row * SROW + c8 * 4);
          f32x4_t hi = *(const f32x4_t*)(st + row * SROW + c8 * 4 + 16);
          if (interior || (m_base + lrow_t < m_end && nv8)) {
            lo += bl;
            hi += bh;
            if (any_res) {
              // the generic epilogue's order, so that a layer computes the same bits on every tile size: (acc + bias) + (r1_hi + r1_lo) + (r2_hi + r2_lo)
              f32x4_t a0, a1, b0, b1;
              widen8<TO>(w1h[it & 3], a0, a1);
              widen8<TO>(w1l[it & 3], b0, b1);
              if (r1) { lo += a0 + b0; hi += a1 + b1; }
              widen8<TO>(w2h[it & 3], a0, a1);
              widen8<TO>(w2l[it & 3], b0, b1);
              if (r2) { lo += a0 + b0; hi += a1 + b1; }
            }
            if constexpr (gelu_k) {
              lo = gelu4<TO>(lo);
              hi = gelu4<TO>(hi);
            } else if (relu) {
              lo = relu4(lo);
              hi = relu4(hi);
            }
            const unsigned eo = ((unsigned)lrow_t * (unsigned)ldo8 + lc8) * 2u;
            i32x4_t ph, pl;
            split8<TO>(lo, hi, ph, pl);
            *(i32x4_t*)(ob + eo) = ph;
            *(i32x4_t*)(ob + eo + olo) = pl;
            if (has_o2) {
              split8<TO>(relu4(lo), relu4(hi), ph, pl);
              *(i32x4_t*)(o2b + eo) = ph;
              *(i32x4_t*)(o2b + eo + olo) = pl;
            }
          }
          if (it + 4 < 8) issue_res(half, it + 4);
          else if (half == 0) issue_res(1, it - 4);
        }
        asm volatile("" ::: "memory");
      }
      stamp_end();
      return;
    }
    f32x4_t bl = {0.f, 0.f, 0.f, 0.f}, bh = bl;
    if (biasp && nv8) {
      bl = *(const f32x4_t*)(biasp + n8);
      bh = *(const f32x4_t*)(biasp + n8 + 4);
    }
    f32x4_t wl = {1.f, 1.f, 1.f, 1.f}, wh = wl;
    if (wsp && nv8) {
      wl = *(const f32x4_t*)(wsp + n8) * p.ascale;
      wh = *(const f32x4_t*)(wsp + n8 + 4) * p.ascale;
    }
    const long trb = (long)m_base * p.ldr + n0;
    const char* q1b = (const char*)p.res1 + trb * 2;
    const char* q2b = (const char*)p.res2 + trb * 2;
    i32x4_t pr1[2][8], pr2[2][8];  // raw 8-element residual vectors
    auto pf8 = [&](int half, int it) __attribute__((always_inline)) {
      if (!any_res) return;
      const int lrow_t = wm * WTM + half * 64 + it * 8 + rsub;
      const bool ok = interior || (m_base + lrow_t < m_end && nv8);
      const unsigned ro = ((unsigned)lrow_t * (unsigned)p.ldr + lc8) * 2u;
      i32x4_t z = {0, 0, 0, 0};
      pr1[half][it] = (r1 && ok) ? *(const i32x4_t*)(q1b + ro) : z;
      pr2[half][it] = (r2 && ok) ? *(const i32x4_t*)(q2b + ro) : z;
    };
#pragma unroll
    for (int it = 0; it < 4; ++it) pf8(0, it);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      asm volatile("" ::: "memory");
      stage_half_f32(half);
      asm volatile("" ::: "memory");
      if (half == 0) {
#pragma unroll
        for (int it = 4; it < 8; ++it) pf8(0, it);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rsub;
        const int lrow_t = wm * WTM + half * 64 + row;
        f32x4_t lo = *(const f32x4_t*)(st + row * SROW + c8 * 4);
        f32x4_t hi = *(const f32x4_t*)(st + row * SROW + c8 * 4 + 16);
        if (interior || (m_base + lrow_t < m_end && nv8)) {
          lo = lo * wl + bl;
          hi = hi * wh + bh;
          if (any_res) {
            f32x4_t a0, a1;
            widen8<TO>(pr1[half][it], a0, a1);
            lo += a0;
            hi += a1;
            widen8<TO>(pr2[half][it], a0, a1);
            lo += a0;
            hi += a1;
          }
          if constexpr (gelu_k) {
            lo = gelu4<TO>(lo);
            hi = gelu4<TO>(hi);
          } else if (relu) {
            lo = relu4(lo);
            hi = relu4(hi);
          }
          const unsigned eo = (unsigned)lrow_t * (unsigned)ldo8 + lc8;
          if (fp8o) {  // e4m3 (saturating) for the next GEMM's A operand
            const float is = p.out_inv_scale;
            auto cl = [&](float a) __attribute__((always_inline)) { return __builtin_amdgcn_fmed3f(a * is, -448.f, 448.f); };
            int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(lo[0]), cl(lo[1]), 0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(lo[2]), cl(lo[3]), w0, true);
            int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(hi[0]), cl(hi[1]), 0, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(hi[2]), cl(hi[3]), w1, true);
            *(long*)(ob + eo) = (long)(unsigned)w0 | ((long)(unsigned)w1 << 32);
          } else if (f32o) {
            *(f32x4_t*)(ob + eo * 4u) = lo;
            *(f32x4_t*)(ob + eo * 4u + 16) = hi;
          } else {
            *(i32x4_t*)(ob + eo * 2u) = pack8<TO>(lo, hi);
          }
          if (has_o2) *(i32x4_t*)(o2b + eo * 2u) = pack8<TO>(relu4(lo), relu4(hi));
        }
        if (half == 0 && it == 3) {
#pragma unroll
          for (int j = 0; j < 4; ++j) pf8(1, j);
        }
      }
      asm volatile("" ::: "memory");
      if (half == 0) {
#pragma unroll
        for (int j = 4; j < 8; ++j) pf8(1, j);
      }
    }
    stamp_end();
    return;
  }
  // ---- fp32-staged forms: read-modify-write residual (EK 1), generic pixel shuffle (EK 3), generic (EK 0) ----
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f}, scale4 = {0.f, 0.f, 0.f, 0.f};
  int ps_co = 0, ps_dy = 0, ps_dx = 0;  // pixel shuffle: the lane's 4 columns fix (tap, channel) once
  if constexpr (pixshuf) {
    const int tap = fdiv(n, p.fd_psC);
    ps_co = n - tap * p.psC;
    ps_dy = p.ps_f == 4 ? tap >> 2 : tap >> 1;
    ps_dx = tap - ps_dy * p.ps_f;
    if (nvalid && biasp) bias4 = *(const f32x4_t*)(biasp + ps_co);
  }
  f32x4_t ws4 = {1.f, 1.f, 1.f, 1.f};  // fp8 operands: dequantisation scale of the lane's 4 columns
  if constexpr (rmw) {
    if (nvalid) {
      if (biasp) bias4 = *(const f32x4_t*)(biasp + n);
      scale4 = *(const f32x4_t*)(MD_SEL_G(p.scale, g) + n);
      if (MD_SEL_G(p.wscale, g)) ws4 = *(const f32x4_t*)(MD_SEL_G(p.wscale, g) + n) * p.ascale;
    }
  }
  // wave-uniform tile base + 32-bit lane offsets (one VGPR per address instead of a 64-bit pair)
  f32x4_t gam4 = {0.f, 0.f, 0.f, 0.f};
  char* ln_b = nullptr;
  if constexpr (ln_emit) {
    if (nvalid) gam4 = *(const f32x4_t*)(MD_SEL_G(p.ln_gamma, g) + n);
    ln_b = (char*)p.ln_out + ((long)m_base * p.ln_ldo + n0) * 2;
  }
  char* out_b = (char*)p.out + (out_boff + (long)m_base * p.ldo + n0) * 4;
  const char* rsrc_b = p.resid_src ? (const char*)p.resid_src + ((long)m_base * p.ldo + n0) * 4 : out_b;  // EPI_RESID_LS: x is read here
  const unsigned lcol = (unsigned)(wn * WTN + col);
  // The read-modify-write epilogue prefetches the fp32 x vectors of a half BEFORE its staging pass, so 16 loads per
  // lane are in flight at once instead of a dependent load -> store chain (that chain, not bandwidth, set the cost of
  // the proj / fc2 epilogues). Register budget: 128 accumulators are live until half 0 is staged: 8 rows of half 0
  // before its staging pass and its other 8 right after it; the first 8 rows of half 1 once rows 0-7 of half 0 have
  // been stored, the last 8 after rows 8-15. Every slot is assigned unconditionally (zeros when out of range).
  f32x4_t pre2[2][16];
  auto prefetch = [&](int half, int it) __attribute__((always_inline)) {
    if constexpr (rmw) {
      const int lrow_t = wm * WTM + half * 64 + (lane >> 4) + it * 4;
      const bool ok = interior || (m_base + lrow_t < m_end && nvalid);
      pre2[half][it] = ok ? *(const f32x4_t*)(rsrc_b + ((unsigned)lrow_t * (unsigned)p.ldo + lcol) * 4u) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  };
#pragma unroll
  for (int it = 0; it < 8; ++it) prefetch(0, it);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int lrow0 = wm * WTM + half * 64 + (lane >> 4);  // tile-local row of iteration 0
    const int m0 = m_base + lrow0;
    asm volatile("" ::: "memory");
    if (!(DIAG && (p.debug_flags & 8))) stage_half_f32(half);
    asm volatile("" ::: "memory");
    if (half == 0) {
#pragma unroll
      for (int it = 8; it < 16; ++it) prefetch(0, it);
    }
    asm volatile("" ::: "memory");
    if constexpr (rmw) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int m = m0 + it * 4;
        const unsigned lr = (unsigned)(lrow0 + it * 4);
        const f32x4_t v = *(const f32x4_t*)(st + row * SROW + col * 4);
        const f32x4_t xnew = resid_ls4(pre2[half][it], scale4, v * ws4 + bias4);
        if (interior || (m < m_end && nvalid)) {
          *(f32x4_t*)(out_b + (lr * (unsigned)p.ldo + lcol) * 4u) = xnew;
#ifndef MD_LNFOLD_NOCOPY
          if constexpr (ln_emit)  // the next GEMM's A operand: round_T(gamma_next . x_new) (split-half: both planes)
#else
          if constexpr (false)
#endif
            store4p<TO>((TO*)(ln_b + (lr * (unsigned)p.ln_ldo + lcol) * 2u), p.ln_plane, xnew * gam4);
        }
#ifndef MD_LNFOLD_NOSTATS  // (timing-only experiment: EXTRA=-DMD_LNFOLD_NOSTATS leaves the statistics out; results are wrong)
        if constexpr (ln_emit) {
          // the row's statistics over this wave's 64 columns: mean, then the centred sum of squares (two 16-lane sums)
          const float mean_w = row_sum16((xnew[0] + xnew[1]) + (xnew[2] + xnew[3])) * (1.0f / 64.0f);
          const f32x4_t dl = xnew - mean_w;
          const float m2_w = row_sum16((dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]));
          if ((lane & 15) == 0) *(f32x2_t*)(lnx + ((int)lr * 4 + wn) * 2) = (f32x2_t){mean_w, m2_w};
        }
#endif
        if (half == 0 && it == 7) {
#pragma unroll
          for (int j = 0; j < 8; ++j) prefetch(1, j);
        }
      }
    } else if constexpr (pixshuf && !is_split<TO>::value) {  // split-half launches reach EK 3 only in its fast form (launch_256)
      const int f = p.ps_f;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int m = m0 + it * 4;
        f32x4_t v = *(const f32x4_t*)(st + row * SROW + col * 4);
        if (m < m_end && nvalid) {
          const int t = fdiv(m, p.fd_psW);
          const int x = m - t * p.psW;
          const int b = fdiv(t, p.fd_psH);
          const int y = t - b * p.psH;
          const long orow = ((long)b * f * p.psH + f * y + ps_dy) * ((long)f * p.psW) + f * x + ps_dx;
          const long o = orow * p.ldo + p.ps_coff + ps_co;
          v += bias4;
          if (p.out_f32)
            store4<float>((float*)p.out + o, v);
          else
            store4<TO>((TO*)p.out + o, v);
          if (p.out2) store4<TO>((TO*)p.out2 + o, relu4(v));
        }
      }
    } else {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int m = m0 + it * 4;
        const f32x4_t v = *(const f32x4_t*)(st + row * SROW + col * 4);
        if (m < m_end && nvalid) epilogue4<TO>(p, g, m, n, v, out_boff);
      }
    }
    asm volatile("" ::: "memory");
    if (half == 0) {
#pragma unroll
      for (int it = 8; it < 16; ++it) prefetch(1, it);
    }
  }
  if constexpr (ln_emit) {
    // the four waves' (mean, M2) of a row -> the tile's: mean_t = their average (64 columns each), M2_t = sum M2_w + 64 sum (mean_w - mean_t)^2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < BM && m_base + tid < m_end) {
      const f32x4_t a = *(const f32x4_t*)(lnx + tid * 8), b = *(const f32x4_t*)(lnx + tid * 8 + 4);
      const float mean_t = ((a[0] + a[2]) + (b[0] + b[2])) * 0.25f;
      const float d0 = a[0] - mean_t, d1 = a[2] - mean_t, d2 = b[0] - mean_t, d3 = b[2] - mean_t;
      const float m2_t = ((a[1] + a[3]) + (b[1] + b[3])) + 64.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
      *(f32x2_t*)(p.ln_stats_out + ((long)(m_base + tid) * p.ln_parts + tile_n) * 2) = (f32x2_t){mean_t, m2_t};
    }
  }
  stamp_end();
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT form of the fc1 GEMM (round 6): dense A, bias (+ LayerNorm fold) + GELU, 2-byte direct stores -- EK 9 / 11 of gemm256_kernel as a
// tile LOOP: one workgroup per CU walks its XCD's share of the raster, and the first k-tile of the NEXT tile is requested before the
// epilogue of the current one, so the 2.5 - 3 us of a tile's prologue (setup, issue, first k-tile landing) pass under the 4 - 5 us of
// the GELU epilogue. What makes that possible here and did not in rounds 1 - 2 (whose tile loop around the staged epilogues spilled
// and lost 5 - 14 %): the direct-store epilogue uses no LDS (the ring is free for the next tile), holds no prefetch arrays, and takes
// bias / c / d and the LayerNorm partials from LDS (requested by LDS-DMA in the last k-tile's load segment), so no compiler-placed
// vmcnt(0) drains the next tile's requests. Stores are younger than those requests: the wait for the next tile's k-tile 0 is
// vmcnt(stores of this epilogue). Same main loop, same arithmetic, same bits as the one-tile kernel.
// LDS exchange area (inside ring slot 4, idle in the last k-tile for KT = 16 / 32): [pairs 2 KB][partials 8 KB][c or bias 1 KB][d 1 KB].
// QKV = true: the fused QKV projection (EPI_QKV, one-plane types): q | k tiles through the lean staged store epilogue, its staging moved to
// ring slots 2 - 3 so that slots 0 - 1 can take the next tile's first k-tile meanwhile; V^T tiles (a third of the launch) staged transposed in
// the OUTPUT type (8.5 KB per wave in slots 2 - 4; the one-tile kernel's fp32 staging covers the whole ring), so they overlap the same way.
// CONV = true (with QKV = true: the plain W image and the lean staged store): the implicit 3 x 3 GEMM with bias (+ ReLU) and one 2-byte
// output -- the first convolution of the decoder's residual units; the tile setup adds the pixel decomposition and the nine taps' border masks.
template <typename T, bool FOLD, bool QKV = false, bool CONV = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, NW = 8, WGN = 4, WTM = 128, WTN = 64, HALF_BYTES = 256 * 128, NSLOT = 5, LPH = 4, KE = 64, PLN = kPlanes<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 3, pc = lane & 7, q16 = lane >> 4, r16 = lane & 15;
  const int lane_off16 = r16 * 128 + ((((r16 >> 1) & 7) ^ q16) << 4);
  const int KT = p.K / KE;
  const long ldw = p.ldw > 0 ? p.ldw : (long)p.K;
  float* const lnx = (float*)(smem + kLnXchg);
  // this workgroup's tiles: XCD x (blocks b, b + 8, .. share one) owns the contiguous id range [xs, xs + xc) of the raster; block j of the
  // XCD takes ids xs + j, xs + j + G / 8, ...
  const int ntiles = p.ptiles, G = gridDim.x;
  int it, xs, xc;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7;
    xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xc = q + (xcd < r ? 1 : 0);
    it = blockIdx.x >> 3;
  }
  const int istep = G >> 3;
  if (it >= xc) return;

  int m_base, m_end, n0, g;
  const char* Wg;
  const char* srcA[LPH];
  unsigned offW[LPH];
  unsigned maskA[LPH];  // CONV: the nine taps' validity per staged row (3 row bits x 3 column bits)
  auto locate = [&](int id) __attribute__((always_inline)) {
    int tile_n, tile_mg;
    if (id < p.map_full_gsz) {
      const int ng = fdiv(id, p.fd_map_gsz), r = id - ng * p.map_gsz;
      tile_mg = fdiv(r, p.fd_map_gn);
      tile_n = ng * p.map_gn + (r - tile_mg * p.map_gn);
    } else {
      const int r = id - p.map_full_gsz;
      tile_mg = fdiv(r, p.fd_map_rn);
      tile_n = p.map_full * p.map_gn + (r - tile_mg * p.map_rn);
    }
    g = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
      if (i < p.ngroups && tile_mg >= p.g_tile0[i]) g = i;
    const int g_row0 = MD_SEL_G(p.g_row0, g), g_arow0 = MD_SEL_G(p.g_arow0, g);
    m_base = g_row0 + (tile_mg - MD_SEL_G(p.g_tile0, g)) * BM;
    m_end = g_row0 + MD_SEL_G(p.g_rows, g);
    n0 = tile_n * BN;
    Wg = (const char*)MD_SEL_G(p.W, g);
    // (the lane's row / chunk indices are re-derived per tile from an opaque copy of the lane id: as loop invariants hipcc keeps a dozen of
    // them alive through the whole tile loop and spills them)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int lrow_o = lane_o >> 3, pc_o = lane_o & 7;
#pragma unroll
    for (int i = 0; i < LPH; ++i) {
      const int r = (i * NW + wave) * 8 + lrow_o;
      const int lc = pc_o ^ ((r >> 1) & 7);
      int m = m_base + r;
      m = m < m_end ? m : m_end - 1;
      const long am = (long)g_arow0 + (m - g_row0);
      if constexpr (CONV) {  // implicit 3 x 3 GEMM (gemm256_kernel's A_CONV3 setup): the centre tap's pixel, the taps' border mask
        const int ow = p.cOW > 0 ? p.cOW : p.cW, oh = p.cOH > 0 ? p.cOH : p.cH;
        const int t2 = fdiv((int)am, p.fd_ow);
        const int ox = (int)am - t2 * ow;
        const int bimg = fdiv(t2, p.fd_oh);
        const int oy = t2 - bimg * oh;
        const int x = ox * p.cstride, y = oy * p.cstride;
        unsigned c3 = 0, mk = 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) c3 |= ((unsigned)(x + kx - 1) < (unsigned)p.cW ? 1u : 0u) << kx;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) mk |= ((unsigned)(y + ky - 1) < (unsigned)p.cH ? c3 : 0u) << (3 * ky);
        maskA[i] = mk;
        const int pix = (bimg * p.cH + y) * p.cW + x;
        srcA[i] = (const char*)p.A + (long)pix * (long)(p.cC * 2) + lc * 16;
      } else {
        maskA[i] = 0;
        srcA[i] = (const char*)p.A + (long)(int)am * (long)(int)(p.lda * 2) + lc * 16;
      }
      const int rp = QKV ? r : ((r & ~63) | (((r >> 5) & 1) << 5) | (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3));  // fc1: the direct-store image
      offW[i] = (unsigned)(n0 + rp) * (unsigned)(ldw * 2) + (unsigned)(lc * 16);
    }
  };
  auto issue_W = [&](int kt, int slot) __attribute__((always_inline)) {
    char* sbase = smem + slot * HALF_BYTES;
    const char* wk = Wg + (long)kt * 128;
#pragma unroll
    for (int i = 0; i < LPH; ++i) glds16(wk + offW[i], sbase + (i * NW + wave) * 1024);
  };
  const int cblocks = CONV ? (p.cCk > 0 ? p.cCk : p.cC) / KE : 1;
  auto issue_A = [&](int kt, int slot) __attribute__((always_inline)) {
    char* sbase = smem + slot * HALF_BYTES;
    if constexpr (CONV) {
      const int tap = fdiv(kt, p.fd_cblocks);
      const int cb = kt - tap * cblocks;
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;  // tap / 3 for tap < 9
      const long a_delta = ((long)(ky - 1) * p.cW + (kx - 1)) * p.cC * 2 + (long)cb * 128;
      int pc_o = lane;
      asm volatile("" : "+v"(pc_o));
      const char* zsrc = (const char*)p.zero_page + (pc_o & 7) * 16;
#pragma unroll
      for (int i = 0; i < LPH; ++i) glds16(((maskA[i] >> tap) & 1u) ? srcA[i] + a_delta : zsrc, sbase + (i * NW + wave) * 1024);
    } else {
      const int kta = (p.a_wrap > 0 && kt >= p.a_wrap) ? kt - p.a_wrap : kt;
#pragma unroll
      for (int i = 0; i < LPH; ++i) glds16(srcA[i] + (long)kta * 128, sbase + (i * NW + wave) * 1024);
    }
  };
  if (p.stagger > 0 && ((blockIdx.x >> 5) & 1)) {  // (gemm256r_kernel's start offset between the halves of an XCD's workgroups: a measurement switch here)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(32);
  }
  // (Measured and not kept, profiles/r06_fc1_loop_ablation.txt: k-tile 1 requested together with k-tile 0 -- the fc1 epilogue leaves ring slots 2 - 3
  // alone -- with the wait at the end of k-tile 0 relaxed to vmcnt(stores + 4) so that it does not wait for the epilogue's store acknowledgements:
  // fc1 +0.3 ms per step; with the plain wait: +-0.)
  locate(xs + it);
  issue_A(0, 0);
  issue_W(0, 1);
  bool first = true, prev_counted = true;  // prev_counted: the previous tile's epilogue issued exactly kStores stores BEHIND this tile's first requests
  constexpr int kStores = 16 * PLN;
  const bool g1 = wm == 1;
  for (;;) {
    f32x4acc_t acc16[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc16[a][b] = (f32x4acc_t){0.f, 0.f, 0.f, 0.f};
    int issued = 2, slot_i = 2, slot_c = 0;
    auto wait_tile = [&](int t) __attribute__((always_inline)) {
      const int younger = issued - (2 * t + 2);
      if (younger >= 3) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else if (younger == 2) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (younger == 1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    // k-tile 0 of this tile: requested before the previous tile's epilogue, whose 16 (x 2 planes) stores are younger -- when that tile was an
    // interior one (every store instruction was issued by every wave); behind a group's last, partial tile a wave may have skipped
    // stores, so everything is waited for
    if (first || !prev_counted) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(MD_PABL) && (MD_PABL & 2)
    } else if (true) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    } else if (kStores == 32) {
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (g1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < KT; ++t) {
      const int slot_w = slot_c == NSLOT - 1 ? 0 : slot_c + 1;
      const char* As = smem + slot_c * HALF_BYTES + wm * WTM * 128;
      const char* Ws = smem + slot_w * HALF_BYTES + wn * WTN * 128;
      slot_c = slot_w == NSLOT - 1 ? 0 : slot_w + 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = lane_off16 ^ (ks << 6);
        i32x4_t wf[4], af[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[a] = *(const i32x4_t*)(Ws + a * 2048 + off);
#pragma unroll
        for (int b = 0; b < 8; ++b) af[b] = *(const i32x4_t*)(As + b * 2048 + off);
        if (t == 0) {  // rest of the pipeline fill, in half-tile order A1 W1 A2 (slots 2, 3, 4)
          if (ks == 0 && KT > 1) { issue_A(1, 2); issue_W(1, 3); issued = 4; slot_i = 4; }
          if (ks == 1 && KT > 2) { issue_A(2, 4); issued = 5; slot_i = 0; }
        } else {
          if (ks == 0 && t + 1 < KT) { issue_W(t + 1, slot_i); ++issued; slot_i = slot_i == NSLOT - 1 ? 0 : slot_i + 1; }
          if (ks == 1 && t + 2 < KT) { issue_A(t + 2, slot_i); ++issued; slot_i = slot_i == NSLOT - 1 ? 0 : slot_i + 1; }
        }
        if (ks == 0 && t == KT - 1) {  // the epilogue's operands into the exchange area (slot 4 is idle in the last k-tile: launch_256 checks KT)
          int l2 = lane;
          asm volatile("" : "+v"(l2));
          if constexpr (FOLD) {
            int r2 = m_base + wave * 32 + (l2 >> 1);
            r2 = r2 < m_end - 1 ? r2 : m_end - 1;
            glds16((const char*)p.ln_stats + (long)r2 * 32 + (l2 & 1) * 16, smem + kLnXchg + 2048 + wave * 1024);
            if (wave == 0) glds16((const char*)(MD_SEL_G(p.ln_c, g) + n0) + l2 * 16, smem + kLnXchg + 10240);
          }
          if (wave == 1) glds16((const char*)(MD_SEL_G(p.bias, g) + n0) + l2 * 16, smem + kLnXchg + 11264);
        }
        if (g1 && ks == 1 && t + 1 < KT) wait_tile(t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) Atom16<T>::mma(wf[a], af[b], acc16[a][b]);
        if (!g1 && ks == 1 && t + 1 < KT) wait_tile(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!g1) __builtin_amdgcn_s_barrier();
    // ---- hand-over: everything of this tile has landed; the next tile's k-tile 0 goes out before the epilogue ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int e_m_base = m_base, e_m_end = m_end, e_n0 = n0;
    it += istep;
    const bool has_next = it < xc;
    __builtin_amdgcn_s_barrier();  // the exchange area is complete and visible; every wave has left the ring
    asm volatile("" ::: "memory");
    const bool e_vt = QKV && !CONV && e_n0 >= 2 * p.embed;  // a V^T tile
    const bool early = has_next;
    if (early) {
      locate(xs + it);
      issue_A(0, 0);
      issue_W(0, 1);
    }
    if constexpr (FOLD) {  // partials -> (rstd, -mu rstd), one row per thread
      if (tid < BM) {
        const f32x4_t p0 = *(const f32x4_t*)(lnx + 512 + tid * 8), p1 = *(const f32x4_t*)(lnx + 512 + tid * 8 + 4);
        const float mu = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
        const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
        const float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + 256.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        const float rstd = 1.0f / sqrtf(m2 * p.ln_inv_n + p.ln_eps);
        *(f32x2_t*)(lnx + 2 * tid) = (f32x2_t){rstd, -mu * rstd};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if constexpr (QKV) {
      const bool interior = e_m_base + BM <= e_m_end;
      int r16e = r16, q16e = q16, lane_e = lane;
      asm volatile("" : "+v"(r16e), "+v"(q16e), "+v"(lane_e));
      if (e_vt) {
        // ---- V^T tile: fold / bias in the ACCUMULATOR layout (the q | k path's form of the same fused multiply-adds), then staged TRANSPOSED in the
        //      output type: 64 n-rows x (64 tokens x 2 B + 8) per wave and half = 8.5 KB per wave in ring slots 2 - 4, below the exchange area -- the
        //      one-tile kernel stages fp32 (17 KB per wave: the whole ring). Slots 0 - 1 take the next tile's first k-tile meanwhile; same values, same
        //      bits, the same 128-byte store segments. ----
        constexpr int SRT = 136;
        char* stt = smem + 2 * HALF_BYTES + wave * (64 * SRT);
        const int hd = (e_n0 - 2 * p.embed + wn * WTN) >> 6;
        f32x4_t bq[4], wq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int cl = wn * WTN + a * 16 + 4 * q16e;
          bq[a] = *(const f32x4_t*)(lnx + 2816 + cl);
          wq[a] = FOLD ? *(const f32x4_t*)(lnx + 2560 + cl) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
        }
        float lnA[8], lnB[8];
        if constexpr (FOLD) {
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            const f32x2_t t2 = *(const f32x2_t*)(lnx + 2 * (wm * WTM + b * 16 + r16e));
            lnA[b] = t2[0];
            lnB[b] = t2[1];
          }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const int b = half * 4 + bb;
              const f32x4acc_t c = acc16[a][b];
              f32x4_t x = {c[0], c[1], c[2], c[3]};
              if constexpr (FOLD) {
                const f32x4_t Bv = {lnB[b], lnB[b], lnB[b], lnB[b]}, Av = {lnA[b], lnA[b], lnA[b], lnA[b]};
                x = fma4(x, Av, fma4(Bv, wq[a], bq[a]));
              } else {
                x = x * wq[a] + bq[a];
              }
              const i32x2_t pk = pack4<T>(x);
              char* wp = stt + (a * 16 + 4 * q16e) * SRT + (bb * 16 + r16e) * 2;
              *(unsigned short*)(wp) = (unsigned short)(pk[0] & 0xffff);
              *(unsigned short*)(wp + SRT) = (unsigned short)((unsigned)pk[0] >> 16);
              *(unsigned short*)(wp + 2 * SRT) = (unsigned short)(pk[1] & 0xffff);
              *(unsigned short*)(wp + 3 * SRT) = (unsigned short)((unsigned)pk[1] >> 16);
            }
          asm volatile("" ::: "memory");
          const int m = e_m_base + wm * WTM + half * 64 + r16e * 4;
          const int seq = fdiv(m, p.fd_seq_stride);
          const int tok = m - seq * p.seq_stride;
          T* vrow = (T*)p.vT + (((long)seq * p.heads + hd) * 64) * p.kpad + tok;
          const bool ok = m < e_m_end;
#pragma unroll
          for (int i2 = 0; i2 < 16; ++i2) {
            const int nl = i2 * 4 + q16e;
            const i32x2_t raw = *(const i32x2_t*)(stt + nl * SRT + r16e * 8);
            if (ok) *(i32x2_t*)(vrow + (long)nl * p.kpad) = raw;
          }
          asm volatile("" ::: "memory");
        }
      } else {
        // ---- q | k tile: the lean staged store epilogue, staging in ring slots 2 - 3 (8 KB per wave) ----
        const float qs = (!CONV && e_n0 < p.embed) ? p.qscale : 1.f;
        const bool relu_o = CONV && p.act == ACT_RELU;
        f32x4_t bq[4], wq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int cl = wn * WTN + a * 16 + 4 * q16e;
          bq[a] = *(const f32x4_t*)(lnx + 2816 + cl) * qs;
          wq[a] = (FOLD ? *(const f32x4_t*)(lnx + 2560 + cl) : (f32x4_t){1.f, 1.f, 1.f, 1.f}) * qs;
        }
        float lnA[8], lnB[8];
        if constexpr (FOLD) {
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            const f32x2_t t2 = *(const f32x2_t*)(lnx + 2 * (wm * WTM + b * 16 + r16e));
            lnA[b] = t2[0] * qs;
            lnB[b] = t2[1];
          }
        }
        char* st = smem + 2 * HALF_BYTES + wave * 8192;
        const long ldo8 = CONV ? (long)p.ldo : 2L * p.embed;
        char* ob = (char*)p.out + ((long)e_m_base * ldo8 + e_n0) * 2;
        const int rsub = lane_e >> 3;
        const unsigned lc8 = (unsigned)(wn * WTN + (lane_e & 7) * 8);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const int b = half * 4 + bb;
              const f32x4acc_t c = acc16[a][b];
              f32x4_t x = {c[0], c[1], c[2], c[3]};
              if constexpr (FOLD) {
                const f32x4_t Bv = {lnB[b], lnB[b], lnB[b], lnB[b]}, Av = {lnA[b], lnA[b], lnA[b], lnA[b]};
                x = fma4(x, Av, fma4(wq[a], Bv, bq[a]));
              } else {
                x = fma4(x, wq[a], bq[a]);
              }
              if (relu_o) x = relu4(x);
              const int row = bb * 16 + r16e;
              const int chunk = (a * 2 + (q16e >> 1)) ^ (row & 7);
              *(i32x2_t*)(st + row * 128 + chunk * 16 + (q16e & 1) * 8) = pack4<T>(x);
            }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i2 = 0; i2 < 8; ++i2) {
            const int row = i2 * 8 + rsub;
            const int lrow_t = wm * WTM + half * 64 + row;
            const i32x4_t raw = *(const i32x4_t*)(st + row * 128 + (((lane_e & 7) ^ (row & 7)) << 4));
            if (interior || e_m_base + lrow_t < e_m_end) *(i32x4_t*)(ob + ((unsigned)lrow_t * (unsigned)ldo8 + lc8) * 2u) = raw;
          }
          asm volatile("" ::: "memory");
        }
      }
    }
    // ---- epilogue: direct stores from the accumulator layout (the permuted W image: gemm256_kernel's EK 9 / 11) ----
    if constexpr (!QKV) {
      const bool interior = e_m_base + BM <= e_m_end;
      // (lane indices made opaque per tile: hipcc otherwise hoists the epilogue's LDS and store offsets out of the TILE loop, spills them and
      // reloads each behind a vmcnt(0) -- which would wait for the next tile's requests)
      int r16e = r16, q16e = q16;
      asm volatile("" : "+v"(r16e), "+v"(q16e));
      f32x4_t bq[4], wq[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int cl = wn * WTN + (a >> 1) * 32 + q16e * 8 + (a & 1) * 4;  // tile-local column of the lane's 4 accumulators of n-block a
        bq[a] = *(const f32x4_t*)(lnx + 2816 + cl);
        wq[a] = FOLD ? *(const f32x4_t*)(lnx + 2560 + cl) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
      }
      float lnA[8], lnB[8];
      if constexpr (FOLD) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const f32x2_t t2 = *(const f32x2_t*)(lnx + 2 * (wm * WTM + b * 16 + r16e));
          lnA[b] = t2[0];
          lnB[b] = t2[1];
        }
      }
      char* ob = (char*)p.out + ((long)e_m_base * p.ldo + e_n0) * 2;
      const unsigned lcd = (unsigned)(wn * WTN + q16e * 8), lo_off = (unsigned)p.o_plane * 2u;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int lrow_t = wm * WTM + b * 16 + r16e;
        const bool ok = interior || e_m_base + lrow_t < e_m_end;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4_t v[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int a = 2 * h + j;
            const f32x4acc_t c = acc16[a][b];
            f32x4_t x = {c[0], c[1], c[2], c[3]};
            if constexpr (FOLD) {
              const f32x4_t Bv = {lnB[b], lnB[b], lnB[b], lnB[b]}, Av = {lnA[b], lnA[b], lnA[b], lnA[b]};
              x = fma4(x, Av, fma4(wq[a], Bv, bq[a]));
            } else {
              x = fma4(x, wq[a], bq[a]);
            }
#if defined(MD_PABL) && (MD_PABL & 1)  // timing-only build: no fold / GELU arithmetic
            v[j] = (f32x4_t){c[0], c[1], c[2], c[3]};
#else
            v[j] = gelu4<T>(x);
#endif
          }
          const unsigned eo = ((unsigned)lrow_t * (unsigned)p.ldo + lcd + (unsigned)(h * 32)) * 2u;
          if constexpr (PLN == 2) {
            i32x4_t ph, pl;
            split8<T>(v[0], v[1], ph, pl);
            if (ok) {
              *(i32x4_t*)(ob + eo) = ph;
              *(i32x4_t*)(ob + eo + lo_off) = pl;
            }
          } else {
            const i32x4_t raw = pack8<T>(v[0], v[1]);
#if defined(MD_PABL) && (MD_PABL & 2)  // timing-only build: no stores (one that never happens keeps the values alive)
            if (ok && raw[0] == 0x7fc07fc1 && raw[3] == 0x12345678) *(i32x4_t*)(ob + eo) = raw;
#else
            if (ok) *(i32x4_t*)(ob + eo) = raw;
#endif
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!has_next) break;
    first = false;
    prev_counted = early && e_m_base + BM <= e_m_end;  // (an interior tile: every wave issued every store instruction)
  }
}

// The read-modify-write GEMMs (proj, fc2: EPI_RESID_LS, with or without the LayerNorm fold's producer part) as the same tile loop. The one-tile
// kernel's fp32 staging (17 KB per wave) covers the whole ring; here a wave stages 32 rows at a time (8.5 KB per wave in ring slots 2 - 4), so
// that slots 0 - 1 can take the next tile's first k-tile under the 12 - 16 us of this epilogue. Same arithmetic, same bits.
// CONVR = true: the implicit 3 x 3 GEMM whose 2-byte store epilogue has residual inputs and / or a relu'd second output (the second convolution of
// the decoder's residual units) on the same skeleton: gemm256_kernel's arithmetic of that form in the 32-row passes.
template <typename T, bool EMIT, bool CONVR = false>
__global__ __launch_bounds__(512) void gemm256r_kernel(const GemmParams p) {
  constexpr bool FOLD = false, QKV = true;  // (QKV = true below only selects the plain W image)
  constexpr int BM = 256, BN = 256, NW = 8, WGN = 4, WTM = 128, WTN = 64, HALF_BYTES = 256 * 128, NSLOT = 5, LPH = 4, KE = 64, PLN = kPlanes<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lrow = lane >> 3, pc = lane & 7, q16 = lane >> 4, r16 = lane & 15;
  const int lane_off16 = r16 * 128 + ((((r16 >> 1) & 7) ^ q16) << 4);
  const int KT = p.K / KE;
  const long ldw = p.ldw > 0 ? p.ldw : (long)p.K;
  float* const lnx = (float*)(smem + kLnXchg);
  // this workgroup's tiles: XCD x (blocks b, b + 8, .. share one) owns the contiguous id range [xs, xs + xc) of the raster; block j of the
  // XCD takes ids xs + j, xs + j + G / 8, ...
  const int ntiles = p.ptiles, G = gridDim.x;
  int it, xs, xc;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7;
    xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xc = q + (xcd < r ? 1 : 0);
    it = blockIdx.x >> 3;
  }
  const int istep = G >> 3;
  if (it >= xc) return;

  int m_base, m_end, n0, g;
  const char* Wg;
  const char* srcA[LPH];
  unsigned offW[LPH];
  unsigned maskA[LPH];  // CONVR: the nine taps' validity per staged row
  auto locate = [&](int id) __attribute__((always_inline)) {
    int tile_n, tile_mg;
    if (id < p.map_full_gsz) {
      const int ng = fdiv(id, p.fd_map_gsz), r = id - ng * p.map_gsz;
      tile_mg = fdiv(r, p.fd_map_gn);
      tile_n = ng * p.map_gn + (r - tile_mg * p.map_gn);
    } else {
      const int r = id - p.map_full_gsz;
      tile_mg = fdiv(r, p.fd_map_rn);
      tile_n = p.map_full * p.map_gn + (r - tile_mg * p.map_rn);
    }
    g = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
      if (i < p.ngroups && tile_mg >= p.g_tile0[i]) g = i;
    const int g_row0 = MD_SEL_G(p.g_row0, g), g_arow0 = MD_SEL_G(p.g_arow0, g);
    m_base = g_row0 + (tile_mg - MD_SEL_G(p.g_tile0, g)) * BM;
    m_end = g_row0 + MD_SEL_G(p.g_rows, g);
    n0 = tile_n * BN;
    Wg = (const char*)MD_SEL_G(p.W, g);
    // (the lane's row / chunk indices are re-derived per tile from an opaque copy of the lane id: as loop invariants hipcc keeps a dozen of
    // them alive through the whole tile loop and spills them)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int lrow_o = lane_o >> 3, pc_o = lane_o & 7;
#pragma unroll
    for (int i = 0; i < LPH; ++i) {
      const int r = (i * NW + wave) * 8 + lrow_o;
      const int lc = pc_o ^ ((r >> 1) & 7);
      int m = m_base + r;
      m = m < m_end ? m : m_end - 1;
      const long am = (long)g_arow0 + (m - g_row0);
      if constexpr (CONVR) {  // implicit 3 x 3 GEMM (gemm256_kernel's A_CONV3 setup): the centre tap's pixel, the taps' border mask
        const int ow = p.cOW > 0 ? p.cOW : p.cW, oh = p.cOH > 0 ? p.cOH : p.cH;
        const int t2 = fdiv((int)am, p.fd_ow);
        const int ox = (int)am - t2 * ow;
        const int bimg = fdiv(t2, p.fd_oh);
        const int oy = t2 - bimg * oh;
        const int x = ox * p.cstride, y = oy * p.cstride;
        unsigned c3 = 0, mk = 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) c3 |= ((unsigned)(x + kx - 1) < (unsigned)p.cW ? 1u : 0u) << kx;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) mk |= ((unsigned)(y + ky - 1) < (unsigned)p.cH ? c3 : 0u) << (3 * ky);
        maskA[i] = mk;
        const int pix = (bimg * p.cH + y) * p.cW + x;
        srcA[i] = (const char*)p.A + (long)pix * (long)(p.cC * 2) + lc * 16;
      } else {
        maskA[i] = 0;
        srcA[i] = (const char*)p.A + (long)(int)am * (long)(int)(p.lda * 2) + lc * 16;
      }
      const int rp = QKV ? r : ((r & ~63) | (((r >> 5) & 1) << 5) | (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3));  // fc1: the direct-store image
      offW[i] = (unsigned)(n0 + rp) * (unsigned)(ldw * 2) + (unsigned)(lc * 16);
    }
  };
  auto issue_W = [&](int kt, int slot) __attribute__((always_inline)) {
    char* sbase = smem + slot * HALF_BYTES;
    const char* wk = Wg + (long)kt * 128;
#pragma unroll
    for (int i = 0; i < LPH; ++i) glds16(wk + offW[i], sbase + (i * NW + wave) * 1024);
  };
  const int cblocks = CONVR ? (p.cCk > 0 ? p.cCk : p.cC) / KE : 1;
  auto issue_A = [&](int kt, int slot) __attribute__((always_inline)) {
    char* sbase = smem + slot * HALF_BYTES;
    if constexpr (CONVR) {
      const int tap = fdiv(kt, p.fd_cblocks);
      const int cb = kt - tap * cblocks;
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;  // tap / 3 for tap < 9
      const long a_delta = ((long)(ky - 1) * p.cW + (kx - 1)) * p.cC * 2 + (long)cb * 128;
      int pc_z = lane;
      asm volatile("" : "+v"(pc_z));
      const char* zsrc = (const char*)p.zero_page + (pc_z & 7) * 16;
#pragma unroll
      for (int i = 0; i < LPH; ++i) glds16(((maskA[i] >> tap) & 1u) ? srcA[i] + a_delta : zsrc, sbase + (i * NW + wave) * 1024);
    } else {
      const int kta = (p.a_wrap > 0 && kt >= p.a_wrap) ? kt - p.a_wrap : kt;
#pragma unroll
      for (int i = 0; i < LPH; ++i) glds16(srcA[i] + (long)kta * 128, sbase + (i * NW + wave) * 1024);
    }
  };
  // start offset between the two halves of every XCD's workgroups (blocks 4 - 7, 12 - 15, .. of the XCD: the four n-tiles of an m-panel stay
  // together): all workgroups run the same tile sequence at the same speed, so without it every epilogue of the launch -- the fp32 residual
  // stream's read + write, HBM-bound -- meets every other one, and the main loops leave HBM idle in between
  if (p.stagger > 0 && ((blockIdx.x >> 5) & 1)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(32);
  }
  locate(xs + it);
  issue_A(0, 0);
  issue_W(0, 1);
  bool first = true, prev_counted = true;  // prev_counted: the previous tile's epilogue issued exactly kStores stores BEHIND this tile's first requests
  const bool g1 = wm == 1;
  for (;;) {
    f32x4acc_t acc16[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc16[a][b] = (f32x4acc_t){0.f, 0.f, 0.f, 0.f};
    int issued = 2, slot_i = 2, slot_c = 0;
    auto wait_tile = [&](int t) __attribute__((always_inline)) {
      const int younger = issued - (2 * t + 2);
      if (younger >= 3) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else if (younger == 2) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (younger == 1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    // k-tile 0 of this tile: requested before the previous tile's epilogue, whose 16 (x 2 planes) stores are younger -- when that tile was an
    // interior one (every store instruction was issued by every wave); behind a group's last, partial tile a wave may have skipped
    // stores, so everything is waited for
#ifdef MD_RABL
    if (true) {
#else
    if (first || !prev_counted) {
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (CONVR) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // (the 16 output stores: a lower bound of what follows the requests, whatever the residual inputs)
    } else {
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // (32: the x stores alone; a safe lower bound of what follows the requests in either form)
    }
    __builtin_amdgcn_s_barrier();
    if (g1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < KT; ++t) {
      const int slot_w = slot_c == NSLOT - 1 ? 0 : slot_c + 1;
      const char* As = smem + slot_c * HALF_BYTES + wm * WTM * 128;
      const char* Ws = smem + slot_w * HALF_BYTES + wn * WTN * 128;
      slot_c = slot_w == NSLOT - 1 ? 0 : slot_w + 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = lane_off16 ^ (ks << 6);
        i32x4_t wf[4], af[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[a] = *(const i32x4_t*)(Ws + a * 2048 + off);
#pragma unroll
        for (int b = 0; b < 8; ++b) af[b] = *(const i32x4_t*)(As + b * 2048 + off);
        if (t == 0) {  // rest of the pipeline fill, in half-tile order A1 W1 A2 (slots 2, 3, 4)
          if (ks == 0 && KT > 1) { issue_A(1, 2); issue_W(1, 3); issued = 4; slot_i = 4; }
          if (ks == 1 && KT > 2) { issue_A(2, 4); issued = 5; slot_i = 0; }
        } else {
          if (ks == 0 && t + 1 < KT) { issue_W(t + 1, slot_i); ++issued; slot_i = slot_i == NSLOT - 1 ? 0 : slot_i + 1; }
          if (ks == 1 && t + 2 < KT) { issue_A(t + 2, slot_i); ++issued; slot_i = slot_i == NSLOT - 1 ? 0 : slot_i + 1; }
        }
        if (ks == 0 && t == KT - 1) {  // the epilogue's column vectors into the exchange area (slot 4 is idle in the last k-tile: launch_256 checks KT)
          int l2 = lane;
          asm volatile("" : "+v"(l2));
          if (wave == 0) glds16((const char*)(MD_SEL_G(p.bias, g) + n0) + l2 * 16, smem + kLnXchg + 8192);
          if constexpr (!CONVR) {
            if (wave == 1) glds16((const char*)(MD_SEL_G(p.scale, g) + n0) + l2 * 16, smem + kLnXchg + 9216);
          }
          if constexpr (EMIT) {
            if (wave == 2) glds16((const char*)(MD_SEL_G(p.ln_gamma, g) + n0) + l2 * 16, smem + kLnXchg + 10240);
          }
        }
        if (g1 && ks == 1 && t + 1 < KT) wait_tile(t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) Atom16<T>::mma(wf[a], af[b], acc16[a][b]);
        if (!g1 && ks == 1 && t + 1 < KT) wait_tile(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!g1) __builtin_amdgcn_s_barrier();
    // ---- hand-over: everything of this tile has landed; the next tile's k-tile 0 goes out before the epilogue ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int e_m_base = m_base, e_m_end = m_end, e_n0 = n0;
    it += istep;
    const bool has_next = it < xc;
    __builtin_amdgcn_s_barrier();  // the exchange area is complete and visible; every wave has left the ring
    asm volatile("" ::: "memory");
    if (has_next) {
      locate(xs + it);
      issue_A(0, 0);
      issue_W(0, 1);
    }
    // ---- epilogue: x(f32) += scale * (acc + bias), in four passes of 32 rows per wave through fp32 staging in ring slots 2 - 4 (8.5 KB per wave:
    //      slots 0 - 1 belong to the next tile's first k-tile); EMIT: + round_T(gamma_next . x_new) and the rows' (mean, M2) (gemm256_kernel's EK 5) ----
    const bool interior = e_m_base + BM <= e_m_end;
    if constexpr (CONVR) {
      // ---- the 2-byte store epilogue with residual inputs and / or a relu'd second output (gemm256_kernel's one-plane form of it: out = act(acc + bias
      //      + res1 + res2), out2 = relu(out)), in four passes of 32 rows per wave through the same fp32 staging; a lane owns 8 columns of 4 rows per
      //      pass, the raw residual vectors of a pass are requested one pass ahead ----
      int r16e = r16, q16e = q16, lane_e = lane;
      asm volatile("" : "+v"(r16e), "+v"(q16e), "+v"(lane_e));
      constexpr int SROW = 272;
      char* st = smem + 2 * HALF_BYTES + wave * (32 * SROW);
      const int c8 = (lane_e & 7) * 8, rsub = lane_e >> 3;
      const f32x4_t bl = *(const f32x4_t*)(lnx + 2048 + wn * WTN + c8), bh = *(const f32x4_t*)(lnx + 2048 + wn * WTN + c8 + 4);
      const bool r1 = p.res1 != nullptr, r2 = p.res2 != nullptr, relu = p.act == ACT_RELU, has_o2 = p.out2 != nullptr;
      const long tb = (long)e_m_base * p.ldo + e_n0, trb = (long)e_m_base * p.ldr + e_n0;
      char* ob = (char*)p.out + tb * 2;
      char* o2b = (char*)p.out2 + tb * 2;
      const char* q1b = (const char*)p.res1 + trb * 2;
      const char* q2b = (const char*)p.res2 + trb * 2;
      const unsigned lc8 = (unsigned)(wn * WTN + c8);
      i32x4_t pr1[4][4], pr2[4][4];
      auto pf8 = [&](int q, int i2) __attribute__((always_inline)) {
        const int lrow_t = wm * WTM + q * 32 + i2 * 8 + rsub;
        const bool ok = interior || e_m_base + lrow_t < e_m_end;
        const unsigned ro = ((unsigned)lrow_t * (unsigned)p.ldr + lc8) * 2u;
        const i32x4_t z = {0, 0, 0, 0};
        pr1[q][i2] = (r1 && ok) ? *(const i32x4_t*)(q1b + ro) : z;
        pr2[q][i2] = (r2 && ok) ? *(const i32x4_t*)(q2b + ro) : z;
      };
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) pf8(0, i2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const f32x4acc_t c = acc16[a][q * 2 + bb];
            *(f32x4_t*)(st + (bb * 16 + r16e) * SROW + (a * 16 + 4 * q16e) * 4) = (f32x4_t){c[0], c[1], c[2], c[3]};
          }
        asm volatile("" ::: "memory");
        if (q < 3) {
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) pf8(q + 1, i2);
        }
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const int row = i2 * 8 + rsub;
          const unsigned lr = (unsigned)(wm * WTM + q * 32 + row);
          f32x4_t lo = *(const f32x4_t*)(st + row * SROW + c8 * 4);
          f32x4_t hi = *(const f32x4_t*)(st + row * SROW + c8 * 4 + 16);
          if (interior || e_m_base + (int)lr < e_m_end) {
            lo = lo * (f32x4_t){1.f, 1.f, 1.f, 1.f} + bl;
            hi = hi * (f32x4_t){1.f, 1.f, 1.f, 1.f} + bh;
            if (r1 || r2) {
              f32x4_t a0, a1;
              widen8<T>(pr1[q][i2], a0, a1);
              lo += a0;
              hi += a1;
              widen8<T>(pr2[q][i2], a0, a1);
              lo += a0;
              hi += a1;
            }
            if (relu) {
              lo = relu4(lo);
              hi = relu4(hi);
            }
            const unsigned eo = (lr * (unsigned)p.ldo + lc8) * 2u;
            *(i32x4_t*)(ob + eo) = pack8<T>(lo, hi);
            if (has_o2) *(i32x4_t*)(o2b + eo) = pack8<T>(relu4(lo), relu4(hi));
          }
        }
        asm volatile("" ::: "memory");
      }
    } else {
      int r16e = r16, q16e = q16, lane_e = lane;
      asm volatile("" : "+v"(r16e), "+v"(q16e), "+v"(lane_e));
      constexpr int SROW = 272;
      char* st = smem + 2 * HALF_BYTES + wave * (32 * SROW);
      const int col = (lane_e & 15) * 4, rq = lane_e >> 4;
      const f32x4_t bias4 = *(const f32x4_t*)(lnx + 2048 + wn * WTN + col);
      const f32x4_t scale4 = *(const f32x4_t*)(lnx + 2304 + wn * WTN + col);
      f32x4_t gam4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (EMIT) gam4 = *(const f32x4_t*)(lnx + 2560 + wn * WTN + col);
      char* out_b = (char*)p.out + ((long)e_m_base * p.ldo + e_n0) * 4;
      char* ln_b = EMIT ? (char*)p.ln_out + ((long)e_m_base * p.ln_ldo + e_n0) * 2 : nullptr;
      const unsigned lcol = (unsigned)(wn * WTN + col);
      f32x4_t pre[4][8];
      auto prefetch = [&](int q, int i2) __attribute__((always_inline)) {
        const int lrow_t = wm * WTM + q * 32 + i2 * 4 + rq;
        const bool ok = interior || e_m_base + lrow_t < e_m_end;
#if defined(MD_RABL) && (MD_RABL & 1)  // timing-only builds (tools/probes/rmw_loop_ablation.sh): no x loads
        pre[q][i2] = (f32x4_t){(float)lrow_t, 0.f, 0.f, 0.f};
#else
        // (non-temporal: the residual stream is read once per GEMM and 0.66 GB at B = 8 -- kept out of the L2's way, proj 12.47 -> 12.03 and fc2
        // 29.48 -> 29.22 ms per step; the same hint on the x stores -0.2 / +0.07, on fc1's stores +1.1 ms: profiles/r06_nt_ab.txt)
        pre[q][i2] = ok ? __builtin_nontemporal_load((const f32x4_t*)(out_b + ((unsigned)lrow_t * (unsigned)p.ldo + lcol) * 4u)) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#endif
      };
#pragma unroll
      for (int i2 = 0; i2 < 8; ++i2) prefetch(0, i2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const f32x4acc_t c = acc16[a][q * 2 + bb];
            *(f32x4_t*)(st + (bb * 16 + r16e) * SROW + (a * 16 + 4 * q16e) * 4) = (f32x4_t){c[0], c[1], c[2], c[3]};
          }
        asm volatile("" ::: "memory");
        if (q < 3) {
#pragma unroll
          for (int i2 = 0; i2 < 8; ++i2) prefetch(q + 1, i2);
        }
#pragma unroll
        for (int i2 = 0; i2 < 8; ++i2) {
          const int row = i2 * 4 + rq;
          const unsigned lr = (unsigned)(wm * WTM + q * 32 + row);
          const f32x4_t v = *(const f32x4_t*)(st + row * SROW + col * 4);
          const f32x4_t xnew = resid_ls4(pre[q][i2], scale4, v + bias4);
#ifdef MD_RABL
          const bool never = xnew[0] == 1.2345678e30f && xnew[3] == -7.7e-30f;  // (keeps the values alive)
          if (interior || e_m_base + (int)lr < e_m_end) {
            if (!(MD_RABL & 2) || never) *(f32x4_t*)(out_b + (lr * (unsigned)p.ldo + lcol) * 4u) = xnew;
            if constexpr (EMIT) {
              if (!(MD_RABL & 4) || never) store4p<T>((T*)(ln_b + (lr * (unsigned)p.ln_ldo + lcol) * 2u), p.ln_plane, xnew * gam4);
            }
          }
          if constexpr (EMIT && !(MD_RABL & 8)) {
#else
          if (interior || e_m_base + (int)lr < e_m_end) {
            *(f32x4_t*)(out_b + (lr * (unsigned)p.ldo + lcol) * 4u) = xnew;
            if constexpr (EMIT) store4p<T>((T*)(ln_b + (lr * (unsigned)p.ln_ldo + lcol) * 2u), p.ln_plane, xnew * gam4);
          }
          if constexpr (EMIT) {
#endif
            const float mean_w = row_sum16((xnew[0] + xnew[1]) + (xnew[2] + xnew[3])) * (1.0f / 64.0f);
            const f32x4_t dl = xnew - mean_w;
            const float m2_w = row_sum16((dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]));
            if ((lane_e & 15) == 0) *(f32x2_t*)(lnx + ((int)lr * 4 + wn) * 2) = (f32x2_t){mean_w, m2_w};
          }
        }
        asm volatile("" ::: "memory");
      }
      if constexpr (EMIT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < BM && e_m_base + tid < e_m_end) {
          const f32x4_t a4 = *(const f32x4_t*)(lnx + tid * 8), b4 = *(const f32x4_t*)(lnx + tid * 8 + 4);
          const float mean_t = ((a4[0] + a4[2]) + (b4[0] + b4[2])) * 0.25f;
          const float d0 = a4[0] - mean_t, d1 = a4[2] - mean_t, d2 = b4[0] - mean_t, d3 = b4[2] - mean_t;
          const float m2_t = ((a4[1] + a4[3]) + (b4[1] + b4[3])) + 64.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
          *(f32x2_t*)(p.ln_stats_out + ((long)(e_m_base + tid) * p.ln_parts + (e_n0 >> 8)) * 2) = (f32x2_t){mean_t, m2_t};
        }
      }
    }
    const bool early = has_next;
    if (!has_next) break;
    first = false;
    prev_counted = early && interior;  // (an interior tile: every wave issued every store instruction)
  }
}


// host side of map_tile: n-tiles are walked in groups of gn (L2-aware raster), the last group may be narrower
static inline void prep_tile_map(GemmParams& p, int tiles_m, int tiles_n) {
  const int gn = p.raster_gn > 0 ? p.raster_gn : tiles_n;
  p.map_gn = gn;
  p.map_gsz = tiles_m * gn;
  p.map_full = tiles_n / gn;
  p.map_full_gsz = p.map_full * p.map_gsz;
  p.map_rn = tiles_n - p.map_full * gn;
  p.fd_map_gsz = make_fastdiv(p.map_gsz);
  p.fd_map_gn = make_fastdiv(gn);
  p.fd_map_rn = make_fastdiv(p.map_rn);
}

template <typename T, int AMODE>
static int launch_256(GemmParams& p, hipStream_t stream) {
  constexpr int BM = 256, BN = 256;
  int tiles_m = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    p.g_tile0[g] = tiles_m;
    tiles_m += cdiv(p.g_rows[g], BM);
  }
  p.g_tile0[p.ngroups] = tiles_m;
  for (int g = p.ngroups + 1; g <= kMaxGroups; ++g) p.g_tile0[g] = tiles_m;
  const long blocks = (long)tiles_m * cdiv(p.N, BN);
  if (blocks <= 0) return MD_OK;
  if (blocks > 0x7fffffffL) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: too many tiles (%ld)", blocks);
  prep_tile_map(p, tiles_m, cdiv(p.N, BN));
  constexpr int smem = 5 * 256 * 128;  // 160 KB: the whole LDS of a CU
  const dim3 grid((unsigned)blocks, (unsigned)(p.batch > 1 ? p.batch : 1));
  auto go = [&](auto kern, std::atomic<unsigned long>* attr_set) -> int {
    // hipFuncAttributeMaxDynamicSharedMemorySize is per DEVICE: one bit per device ordinal (a process may drive several GPUs)
    int ordinal = 0;
    MD_HIP(hipGetDevice(&ordinal));
    const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
    if (!bit || !(attr_set->load(std::memory_order_acquire) & bit)) {  // concurrent first launches at worst both set the attribute
      MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      attr_set->fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, p);
    MD_HIP(hipGetLastError());
    return MD_OK;
  };
  // epilogue specialisation (EK) from the runtime parameters
  int ek = 0;
  const bool ln_prod = p.ln_out != nullptr, ln_cons = p.ln_stats != nullptr;
  if (ln_prod || ln_cons) {  // the LayerNorm fold exists for the dense 16-bit forms of the 256 x 256 kernel only: refuse anything else loudly
    if (AMODE != A_DENSE || sizeof(typename OutT<T>::type) != 2 || std::is_same<T, fp8_t>::value || p.N % BN != 0 || p.batch > 1)
      MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: the LayerNorm fold needs dense 16-bit operands and N %% 256 == 0 (N=%d)", p.N);
    if (ln_prod && (p.epi != EPI_RESID_LS || !p.ln_stats_out || p.ln_parts != p.N / BN || !p.ln_gamma[0] || p.ln_ldo % 4 != 0))
      MD_FAIL(MD_ERR_INVALID_ARG, "gemm: LayerNorm-fold producer parameters");
    if (ln_cons && (!p.ln_c[0] || !p.bias[0] || p.wscale[0] || p.res1 || p.res2 || p.out2 || p.out_f32 || p.out_fp8 ||
                    !((p.epi == EPI_QKV && p.embed % BN == 0) || (p.epi == EPI_STORE && p.act == ACT_GELU && p.res_mod == 0))))
      MD_FAIL(MD_ERR_INVALID_ARG, "gemm: LayerNorm-fold consumer parameters");
  }
  if (p.epi == EPI_RESID_LS) ek = ln_prod ? 5 : 1;
  else if (p.epi == EPI_PIXSHUF) ek = 3;
  else if (sizeof(typename OutT<T>::type) == 2) {
    const long ldo_e = p.epi == EPI_QKV ? 2L * p.embed : p.ldo;
    const bool vec8 = p.N % 8 == 0 && ldo_e % 8 == 0 && (!(p.res1 || p.res2) || p.ldr % 8 == 0);
    // split-half outputs take the store kinds only for their lean sub-path (no residual inputs, no second / fp32 output) and,
    // for q | k tiles, when a tile cannot straddle the q | k sections of the [q_hi | q_lo | k_hi | k_lo] rows
    // (round 6: residual inputs and the relu'd second output included)
    const bool split_ok = !is_split<T>::value || (!p.out_f32 && !p.out_fp8 && p.o_plane % 8 == 0 && (!(p.res1 || p.res2) || p.r_plane % 8 == 0) &&
                                                  (p.epi != EPI_QKV || p.embed % BN == 0));
    if (!split_ok) ek = 0;
    else if (vec8 && p.epi == EPI_STORE && p.res_mod == 0 && p.act == ACT_GELU) ek = ln_cons ? 7 : 4;
    else if (vec8 && ((p.epi == EPI_STORE && p.res_mod == 0 && p.act != ACT_GELU) || (p.epi == EPI_QKV && (2 * p.embed) % BN == 0))) ek = (ln_cons && p.epi == EPI_QKV) ? 6 : 2;
    if (ln_cons && ek != 6 && ek != 7) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: the LayerNorm-fold consumer needs the 16-byte store epilogue (N, ldo %% 8 == 0)");
  }
  const bool diag = p.stamps != nullptr || p.debug_flags != 0;
  static std::atomic<unsigned long> set[12], dset[5];  // zero-initialised statics
  // the direct-store form of a lean store launch (GemmParams::direct_store)
  const bool lean = !p.res1 && !p.res2 && !p.out2 && !p.out_f32 && !p.out_fp8 && p.N % BN == 0 && p.batch <= 1;
  if constexpr (sizeof(typename OutT<T>::type) == 2) {
    // Measured in the model (bench.py --direct-store off | on, one box, twice each; profiles/r06_direct_store_ab.txt): the GELU kinds gain
    // (fc1 36.18 -> 35.46 ms per step: their stores leave spread out between the polynomial's arithmetic), the plain kinds LOSE (qkv 23.61 ->
    // 24.52: sixteen half-line stores per wave in one burst cost more than the LDS transpose they replace) -- so only EK 4 / 7 take the
    // direct form; EK 8 / 10 stay instantiable for A/B builds (MD_DIRECT_STORE_ALL).
    if constexpr (AMODE == A_CONV3 && !is_split<T>::value && !std::is_same<T, fp8_t>::value) {
      // the implicit 3 x 3 GEMM with a lean store epilogue (bias, optional ReLU, one 2-byte output: the first convolution of the decoder's residual
      // units) as the tile loop of the q | k form (persist bit 8)
      const int KTc = p.K / 64;
      if (!diag && (p.persist & 8) && lean && ek == 2 && p.epi == EPI_STORE && !p.wscale[0] && p.bias[0] && p.ngroups <= 1 && p.a_wrap == 0 && blocks >= 1024 &&
          KTc >= 3 && (2 * KTc - 2) % 5 != 4 && (2 * KTc - 1) % 5 != 4 && p.N == BN && (p.act == ACT_NONE || p.act == ACT_RELU)) {
        int ordinal = 0, cus = 0;
        MD_HIP(hipGetDevice(&ordinal));
        MD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ordinal));
        const int G = cus >= 8 ? (cus & ~7) : 8;
        p.ptiles = (int)blocks;
        p.stagger = 0;
        static std::atomic<unsigned long> cset;
        const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
        auto kern = gemm256p_kernel<T, false, true, true>;
        if (!bit || !(cset.load(std::memory_order_acquire) & bit)) {
          MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
          cset.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), smem, stream, p);
        MD_HIP(hipGetLastError());
        return MD_OK;
      }
      // the same with residual inputs and / or a relu'd second output (the second convolution of a residual unit: x + conv2(..) [+ skip], and
      // its relu'd copy for the next unit): the read-modify-write loop's skeleton, 32-row staging passes
      if (!diag && (p.persist & 8) && !lean && ek == 2 && p.epi == EPI_STORE && (p.res1 || p.res2 || p.out2) && !p.out_f32 && !p.out_fp8 && !p.wscale[0] &&
          p.bias[0] && p.ngroups <= 1 && p.a_wrap == 0 && p.batch <= 1 && blocks >= 1024 && KTc >= 3 && (2 * KTc - 2) % 5 != 4 && (2 * KTc - 1) % 5 != 4 &&
          p.N == BN && (p.act == ACT_NONE || p.act == ACT_RELU)) {
        int ordinal = 0, cus = 0;
        MD_HIP(hipGetDevice(&ordinal));
        MD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ordinal));
        const int G = cus >= 8 ? (cus & ~7) : 8;
        p.ptiles = (int)blocks;
        p.stagger = 0;
        static std::atomic<unsigned long> c2set;
        const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
        auto kern = gemm256r_kernel<T, false, true>;
        if (!bit || !(c2set.load(std::memory_order_acquire) & bit)) {
          MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
          c2set.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), smem, stream, p);
        MD_HIP(hipGetLastError());
        return MD_OK;
      }
    }
    if constexpr (AMODE == A_DENSE && !std::is_same<T, fp8_t>::value) {
      // the persistent tile loop of the fc1 form (gemm256p_kernel): GemmParams::persist, enough tiles to give every CU several, a k-tile
      // count that leaves ring slot 4 idle in the last k-tile (16, 32: K' = 1024, 2048)
      const int KTp = p.K / 64;
      if (!diag && (p.persist & 1) && p.direct_store && lean && (ek == 4 || ek == 7) && !p.wscale[0] && p.bias[0] && blocks >= 1024 && KTp >= 3 &&
          (2 * KTp - 2) % 5 != 4 && (2 * KTp - 1) % 5 != 4 && (ek == 4 || p.ln_raw)) {
        int ordinal = 0, cus = 0;
        MD_HIP(hipGetDevice(&ordinal));
        MD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ordinal));
        const int G = cus >= 8 ? (cus & ~7) : 8;  // (a multiple of 8: XCD x = blocks x, x + 8, ..; more workgroups than CUs only costs residency)
        p.ptiles = (int)blocks;
        p.stagger = blocks >= 2048 ? gemm_stagger_ticks(2) : 0;
        auto gop = [&](auto kern, std::atomic<unsigned long>* attr_set) -> int {
          const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
          if (!bit || !(attr_set->load(std::memory_order_acquire) & bit)) {
            MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set->fetch_or(bit, std::memory_order_release);
          }
          hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), smem, stream, p);
          MD_HIP(hipGetLastError());
          return MD_OK;
        };
        static std::atomic<unsigned long> pset[2];
        if (ek == 7) return gop(gemm256p_kernel<T, true>, &pset[1]);
        return gop(gemm256p_kernel<T, false>, &pset[0]);
      }
      // the read-modify-write GEMMs (proj, fc2; with the LayerNorm fold's producer part: EK 5) as the tile loop gemm256r_kernel
      if (!diag && (p.persist & 4) && (ek == 1 || ek == 5) && !p.wscale[0] && p.bias[0] && p.scale[0] && !p.resid_src && p.N % BN == 0 && p.batch <= 1 &&
          blocks >= 1024 && KTp >= 3 && (2 * KTp - 2) % 5 != 4 && (2 * KTp - 1) % 5 != 4) {
        int ordinal = 0, cus = 0;
        MD_HIP(hipGetDevice(&ordinal));
        MD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ordinal));
        const int G = cus >= 8 ? (cus & ~7) : 8;  // (a multiple of 8: XCD x = blocks x, x + 8, ..; more workgroups than CUs only costs residency)
        p.ptiles = (int)blocks;
        p.stagger = blocks >= 2048 ? gemm_stagger_ticks(KTp <= 16 ? 0 : 1) : 0;  // (the offset is idle time at either end of the launch: 8 rounds of tiles and more)
        auto gop = [&](auto kern, std::atomic<unsigned long>* attr_set) -> int {
          const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
          if (!bit || !(attr_set->load(std::memory_order_acquire) & bit)) {
            MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set->fetch_or(bit, std::memory_order_release);
          }
          hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), smem, stream, p);
          MD_HIP(hipGetLastError());
          return MD_OK;
        };
        static std::atomic<unsigned long> rset[2];
        if (ek == 5) return gop(gemm256r_kernel<T, true>, &rset[1]);
        return gop(gemm256r_kernel<T, false>, &rset[0]);
      }
      // the fused QKV projection as the same tile loop (one-plane types)
      if constexpr (!is_split<T>::value) {
        if (!diag && (p.persist & 2) && lean && p.epi == EPI_QKV && (ek == 2 || ek == 6) && !p.wscale[0] && p.bias[0] && !p.qkn_g[0] && blocks >= 768 &&
            KTp >= 3 && (2 * KTp - 2) % 5 != 4 && (2 * KTp - 1) % 5 != 4 && (ek == 2 || p.ln_raw) && p.embed % BN == 0 && (p.seq_stride & 3) == 0 &&
            p.N == 3 * p.embed) {
          int ordinal = 0, cus = 0;
          MD_HIP(hipGetDevice(&ordinal));
          MD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ordinal));
          const int G = cus >= 8 ? (cus & ~7) : 8;  // (a multiple of 8: XCD x = blocks x, x + 8, ..; more workgroups than CUs only costs residency)
          p.ptiles = (int)blocks;
          p.stagger = blocks >= 2048 ? gemm_stagger_ticks(3) : 0;
          auto gop = [&](auto kern, std::atomic<unsigned long>* attr_set) -> int {
            const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
            if (!bit || !(attr_set->load(std::memory_order_acquire) & bit)) {
              MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
              attr_set->fetch_or(bit, std::memory_order_release);
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), smem, stream, p);
            MD_HIP(hipGetLastError());
            return MD_OK;
          };
          static std::atomic<unsigned long> qset[2];
          if (ek == 6) return gop(gemm256p_kernel<T, true, true>, &qset[1]);
          return gop(gemm256p_kernel<T, false, true>, &qset[0]);
        }
      }
    }
    if (!diag && p.direct_store && lean) {
#ifdef MD_DIRECT_STORE_ALL
      if constexpr (AMODE != A_CONV3) {
        if (ek == 2) return go(gemm256_kernel<T, AMODE, 8, false>, &set[8]);
      }
#endif
      if constexpr (AMODE == A_DENSE) {
        if (ek == 4) return go(gemm256_kernel<T, AMODE, 9, false>, &set[9]);
        if constexpr (!std::is_same<T, fp8_t>::value) {
#ifdef MD_DIRECT_STORE_ALL
          if (ek == 6) return go(gemm256_kernel<T, AMODE, 10, false>, &set[10]);
#endif
          if (ek == 7) return go(gemm256_kernel<T, AMODE, 11, false>, &set[11]);
        }
      }
    }
  }
  if constexpr (AMODE == A_DENSE && sizeof(typename OutT<T>::type) == 2 && !std::is_same<T, fp8_t>::value) {
    if (ek == 5) return go(gemm256_kernel<T, AMODE, 5, false>, &set[5]);
    if (ek == 6) return go(gemm256_kernel<T, AMODE, 6, false>, &set[6]);
    if (ek == 7) return go(gemm256_kernel<T, AMODE, 7, false>, &set[7]);
  }
  if (diag) {  // md_bench_gemm only: the stamped / ablation build exists for dense bf16 operands
    if constexpr (std::is_same<T, bf16_t>::value && AMODE == A_DENSE) {
      switch (ek) {
        case 1: return go(gemm256_kernel<T, AMODE, 1, true>, &dset[1]);
        case 2: return go(gemm256_kernel<T, AMODE, 2, true>, &dset[2]);
        case 3: return go(gemm256_kernel<T, AMODE, 3, true>, &dset[3]);
        case 4: return go(gemm256_kernel<T, AMODE, 4, true>, &dset[4]);
        default: return go(gemm256_kernel<T, AMODE, 0, true>, &dset[0]);
      }
    } else {
      MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: the diagnostic build (stamps / ablation flags) exists for dense bf16 operands only");
    }
  }
  (void)dset;
  switch (ek) {
    case 1: return go(gemm256_kernel<T, AMODE, 1, false>, &set[1]);
    case 3:  // split-half operands: the fast form only (16-byte stores of both planes); anything else takes the generic kind
      if (!is_split<T>::value || p.ps_fast) return go(gemm256_kernel<T, AMODE, 3, false>, &set[3]);
      break;
    case 2:
      if constexpr (sizeof(typename OutT<T>::type) == 2) return go(gemm256_kernel<T, AMODE, 2, false>, &set[2]);
      break;
    case 4:  // the GELU store kind is built for dense A only (the MLP's fc1); a GELU behind a gathered / convolution A operand
             // takes the store kind's runtime activation path of the generic epilogue
      if constexpr (sizeof(typename OutT<T>::type) == 2 && AMODE == A_DENSE) return go(gemm256_kernel<T, AMODE, 4, false>, &set[4]);
      break;
    default: break;
  }
  return go(gemm256_kernel<T, AMODE, 0, false>, &set[0]);
}

// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WGM, int WGN, int AMODE, int KSPLIT = 1>
static int launch_cfg(GemmParams& p, hipStream_t stream) {
  int tiles_m = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    p.g_tile0[g] = tiles_m;
    tiles_m += cdiv(p.g_rows[g], BM);
  }
  p.g_tile0[p.ngroups] = tiles_m;
  for (int g = p.ngroups + 1; g <= kMaxGroups; ++g) p.g_tile0[g] = tiles_m;
  const int tiles_n = cdiv(p.N, BN);
  const long blocks = (long)tiles_m * tiles_n;
  if (blocks <= 0) return MD_OK;
  if (blocks > 0x7fffffffL) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: too many tiles (%ld)", blocks);
  prep_tile_map(p, tiles_m, tiles_n);
  constexpr int smem = 2 * (BM + BN) * 128 * KSPLIT;
  auto kern = gemm_kernel<T, BM, BN, WGM, WGN, AMODE, KSPLIT>;
  static std::atomic<unsigned long> attr_set{0};  // one bit per device ordinal (the attribute is per device)
  {
    int ordinal = 0;
    MD_HIP(hipGetDevice(&ordinal));
    const unsigned long bit = (ordinal >= 0 && ordinal < 64) ? 1ul << ordinal : 0ul;
    if (!bit || !(attr_set.load(std::memory_order_acquire) & bit)) {
      MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      attr_set.fetch_or(bit, std::memory_order_release);
    }
  }
  if (KSPLIT > 1) gemm_count_ksplit_launch();
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(WGM * WGN * 64 * KSPLIT), smem, stream, p);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// k-groups for a 64 x 64 launch (gemm_kernel's KSPLIT): only launches of at most kSplitBlocks workgroups (one or two four-wave
// workgroups per CU: a wave or two per SIMD, each paying a full L2 round trip per k-tile) behind a long dependent k-loop, only
// 16-bit operand types (the fp32 mode keeps one summation order for every launch size). 160 -> 512: DA3 metric_large at 518^2
// (proj / fc2: 352 workgroups) 206 -> 221 frames/s, config 2 and config 5 unchanged.
constexpr int kSplitBlocks = 512;
template <typename T>
static int pick_ksplit(const GemmParams& p) {
  if (!p.ksplit_ok || std::is_same<T, float>::value || std::is_same<T, fp8_t>::value || p.batch > 1) return 1;
  if (p.epi == EPI_HEAD || p.epi == EPI_HEAD_UP2 || p.qkn_g[0]) return 1;  // (the fused q/k-norm epilogue has workgroup barriers of its own)
  long tiles_m = 0;
  for (int g = 0; g < p.ngroups; ++g) tiles_m += cdiv(p.g_rows[g], 64);
  if (tiles_m * cdiv(p.N, 64) > kSplitBlocks) return 1;
  const int KT = p.K / (128 / (int)sizeof(T));
  for (int ks = 4; ks >= 2; --ks)  // the most groups that divide the k-tiles and leave each at least five of them
    if (KT % ks == 0 && KT / ks >= 5) return ks;
  return 1;
}

template <typename T, int AMODE>
static int launch_tile(GemmParams& p, int tile, hipStream_t stream) {
  switch (tile) {
    case TILE_256x256:
      return launch_256<T, AMODE>(p, stream);
    case TILE_128x128:
      return launch_cfg<T, 128, 128, 2, 2, AMODE>(p, stream);
    case TILE_256x32:
      return launch_cfg<T, 256, 32, 8, 1, AMODE>(p, stream);
    case TILE_128x64:  // N <= 64 (the 64-feature DPT head) and launches that leave CUs idle on 128^2 tiles
      return launch_cfg<T, 128, 64, 2, 2, AMODE>(p, stream);
    case TILE_64x64:
      if constexpr (!std::is_same<T, float>::value && !std::is_same<T, fp8_t>::value) {
        const int ks = pick_ksplit<T>(p);
        if (ks == 4) return launch_cfg<T, 64, 64, 2, 2, AMODE, 4>(p, stream);
        if (ks == 3) return launch_cfg<T, 64, 64, 2, 2, AMODE, 3>(p, stream);
        if (ks == 2) return launch_cfg<T, 64, 64, 2, 2, AMODE, 2>(p, stream);
      }
      return launch_cfg<T, 64, 64, 2, 2, AMODE>(p, stream);
    default:
      MD_FAIL(MD_ERR_INVALID_ARG, "gemm: unknown tile config %d", tile);
  }
}


template <typename T>
static int launch_gemm_typed(GemmParams& p, int amode, int tile, hipStream_t stream) {
  switch (amode) {
    case A_DENSE:
      return launch_tile<T, A_DENSE>(p, tile, stream);
    case A_INDEXED:
      return launch_tile<T, A_INDEXED>(p, tile, stream);
    case A_CONV3:
      return launch_tile<T, A_CONV3>(p, tile, stream);
    default:
      MD_FAIL(MD_ERR_INVALID_ARG, "gemm: bad amode %d", amode);
  }
}

}  // namespace md
