// Ceiling probe for the fused attention kernel (gfx950): the per-tile INSTRUCTION MIX of attention_kernel's fast body --
// 8 + 8 v_mfma_f32_32x32x16_bf16 (S^T = K.Q^T over 4 k-steps for two 32-key blocks; O^T += V^T.P^T for 2 x 2 x 2 steps),
// 32 v_exp_f32, 32 + 3 row-sum adds, 16 v_cvt_pk_bf16_f32 -- with the kernel's data dependencies (the S chain of four MFMAs per
// block, exponentials on its accumulator, the packed exponentials as the B operand of the P.V MFMAs), on random bf16 operands,
// with the kernel's memory side added one piece at a time:
//   0 full mix, fragments in registers      1 MFMAs only      2 no exponentials      3 no row sums
//   4 + 16 ds_read_b128 fragment reads per tile from a static LDS image (read before use)      5 the same, whole tile in flight
//   6 + 4 LDS-DMA per wave and tile (issue only)      7 + s_waitcnt vmcnt(0) + s_barrier per tile, L2-resident source
//   8 strided K / V^T rows (L2 hits after the first pass)      9 every workgroup streams its own tiles from HBM (HBM-bound)
//   10 five workgroups share a (sequence, head): the kernel's traffic      11 the same with a three-stage ring
// each at 1 .. 4 workgroups of 4 waves per CU. What it measures: the rate hipcc's schedule of this structure reaches; the shipped
// kernel sits at row 10's rate (profiles/r03_attention_mix_ceiling.txt).
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize [-DM16=1] -o /tmp/attn_mix_probe tools/probes/attn_mix_probe.hip
//        (-DM16=1: the same FLOPs on v_mfma_f32_16x16x32_bf16, two per 32x32x16, one A fragment per two B fragments)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;
#ifndef M16
#define M16 0
#endif
#ifndef DOT2
#define DOT2 0  // 1: row sums as 16 v_dot2c_f32_bf16 on the packed P words (sums the ROUNDED p) instead of 32 v_add_f32
#endif

// M16 = 1: the FLOPs of one 32x32x16 as two v_mfma_f32_16x16x32_bf16 on two of the four 4-register quarters of the accumulator
// (pair `h` = quarters 2h, 2h+1): a 16 x 16 tiling of the same S^T / O^T blocks -- every K / V^T fragment feeds the two 16-query
// tiles, an S quarter sees two dependent MFMAs per tile instead of four on the whole block. Layout-agnostic (a timing probe); the
// chip holds a higher clock on this shape (MI355X_MICROARCH.md).
__device__ __forceinline__ f16v mma(const i4& a, const i4& b, const f16v& c, int h = 0, const i4* b2p = nullptr) {
#if M16
  const i4 b2 = b2p ? *b2p : b;  // the second 16-query tile has its own B fragment (otherwise the two MFMAs are one expression)
  f16v r = c;
#pragma unroll
  for (int qd = 0; qd < 2; ++qd) {
    const int o = 8 * h + 4 * qd;
    f4v t = {r[o], r[o + 1], r[o + 2], r[o + 3]};
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, qd ? b2 : b), t, 0, 0, 0);
    r[o] = t[0]; r[o + 1] = t[1]; r[o + 2] = t[2]; r[o + 3] = t[3];
  }
  return r;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ int pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  return __builtin_bit_cast(int, __builtin_convertvector((f2){a, b}, bf2));
}

template <int VARIANT, int WPS>
__global__ __launch_bounds__(256, WPS) void mix_kernel(const int* __restrict__ seed, float* __restrict__ out, int tiles, const char* __restrict__ big) {
  const int lane = threadIdx.x & 63;
  // variants 4 / 5: the sixteen fragments of a tile come from a 16-KB LDS image (conflict-free ds_read_b128, the kernel's
  // swizzle), still without LDS-DMA and barrier: 4 = each group of four reads right before its MFMAs (what the kernel's source
  // says), 5 = the eight K reads before the S MFMAs and the eight V^T reads behind them (whole tile in flight)
  // variants 6 / 7: variant 4 + the kernel's streaming -- a two-stage ring, four LDS-DMA wave-instructions per wave and tile
  // (buffer_load ... lds, 1 KB each) from a 1-MB global table: 6 = issue only (no wait: what the issue itself costs), 7 = with
  // s_waitcnt vmcnt(0) + s_barrier at the top of every tile (the kernel's `top`)
  constexpr bool LDSV = VARIANT >= 4;
  constexpr bool STREAM = VARIANT >= 6;
  constexpr bool FARSRC = VARIANT >= 8;  // 8: the kernel's strided rows from a 512-MB buffer; 9: contiguous tiles from it
  // variants 10 / 11: the kernel's sharing -- five consecutive workgroups are the q blocks of ONE (sequence, head) unit and stream the
  // same K / V^T tiles (N = 4096 keys, q|k rows of 4096 B with the head's 128 B inside, V^T rows of 8192 B), every tile new to the L2
  // the first time one of the five asks for it: 10 = two stages (tile t+1 requested at the top of tile t: the kernel), 11 = three
  // stages (tile t+2 requested at the top of tile t, s_waitcnt vmcnt(4))
  constexpr bool SHARED = VARIANT >= 10;
  constexpr int NST = VARIANT == 11 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char tile_img[LDSV ? (STREAM ? NST * 16384 : 16384) : 16];
  if (LDSV) {
    for (int i = threadIdx.x; i < 1024; i += 256) ((i4*)tile_img)[i] = *(const i4*)(seed + (i & 1023) * 4);
    __syncthreads();
  }
  const int hh = lane >> 5, cc = lane & 31;
  int koffp[2], voffp[2];
  for (int sub = 0; sub < 2; ++sub) {
    const int R = sub * 32 + cc;
    koffp[sub] = R * 128 + ((((R >> 1) & 7) ^ hh) << 4);
    voffp[sub] = 8192 + R * 128 + ((((R >> 1) & 7) ^ hh) << 4);
  }
  // loop-invariant "fragments" (opaque to the compiler: loaded from memory once)
  // (one K fragment per 32-key block and one V^T fragment per d-tile, reused over the k-steps: the real kernel reads them from
  // LDS just in time, so they must not occupy 64 registers here -- the probe has to fit four waves per SIMD like the kernel)
  i4 kf[2], vf[2], qf[4];
  for (int s = 0; s < 4; ++s) qf[s] = *(const i4*)(seed + ((lane + s * 64) & 1023) * 4);
  for (int sub = 0; sub < 2; ++sub) kf[sub] = *(const i4*)(seed + ((lane + 256 + sub * 64) & 1023) * 4);
  for (int dt = 0; dt < 2; ++dt) vf[dt] = *(const i4*)(seed + ((lane + 512 + dt * 64) & 1023) * 4);
  f16v o[2] = {{0.f}, {0.f}};
  float l_run = 0.f;
  const int wave = threadIdx.x >> 6;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)(FARSRC ? (const void*)big : (const void*)seed), 0, FARSRC ? (1 << 29) : (1 << 20), 0x00020000);
  // 8: lane (r = lane >> 3, chunk = lane & 7) of DMA i reads 16 B of K row (8 wave + r + 32 i) at a 4096-B row stride (i < 2) or of
  //    V^T row at a 1280-B stride (i >= 2); a tile advances K by 64 rows, V^T by 128 B; each workgroup has its own 2-MB region
  const int wg_base = (int)((blockIdx.x * 2654435761u) & 0xff) * (1 << 21);
  const int kv_k = wg_base + ((wave * 8 + (lane >> 3)) * 4096 + (lane & 7) * 16);
  const int kv_v = wg_base + (1 << 20) + ((wave * 8 + (lane >> 3)) * 1280 + (lane & 7) * 16);
  // the kernel's XCD-aware id map: blocks b and b + 8 share an XCD, each XCD walks a contiguous id range -- the five sharers of a
  // unit run on ONE XCD (one L2) at about the same time
  int lid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int unit = lid / 5, useq = unit >> 4, uhead = unit & 15;
  const int sh_k = useq * (24 << 20) + 2048 + uhead * 128 + (wave * 8 + (lane >> 3)) * 4096 + (lane & 7) * 16;
  const int sh_v = useq * (24 << 20) + (16 << 20) + uhead * (512 << 10) + (wave * 8 + (lane >> 3)) * 8192 + (lane & 7) * 16;
  const int voff_dma = ((lane >> 3) * 128 + (lane & 7) * 16) + wave * 1024 + (blockIdx.x & 15) * 16384;
  for (int t = 0; t < tiles; ++t) {
    if (LDSV) asm volatile("" ::: "memory");  // the image counts as rewritten every tile: no hoisting of the fragment reads
    if constexpr (STREAM) {
      constexpr int DIST = NST - 1;  // tiles of prefetch distance
      auto dma = [&](int tt) __attribute__((always_inline)) {  // requests tile tt into its ring slot
        __attribute__((address_space(3))) char* sb = (__attribute__((address_space(3))) char*)(tile_img + (tt % NST) * 16384 + wave * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (SHARED) {
            const int t6 = tt & 63;
            if (i < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, sb + i * 4096, 16, sh_k, (t6 * 64 + i * 32) * 4096, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, sb + i * 4096, 16, sh_v, t6 * 128 + (i - 2) * 32 * 8192, 0, 0);
          } else if constexpr (VARIANT == 8) {
            const int t4 = tt & 3;  // four tiles per region, then again (L2 hits after the first pass)
            if (i < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, sb + i * 4096, 16, kv_k, (t4 * 64 + i * 32) * 4096, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, sb + i * 4096, 16, kv_v, t4 * 128 + (i - 2) * 32 * 1280, 0, 0);
          } else if constexpr (VARIANT == 9) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, sb + i * 4096, 16, voff_dma + wg_base, ((tt & 63) * 16384 + i * 4096), 0, 0);
          } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, sb + i * 4096, 16, voff_dma, ((tt & 31) * 16384 + i * 4096) & 0xfffff, 0, 0);
          }
        }
      };
      if (t == 0 && DIST == 2) dma(1);
      if constexpr (VARIANT >= 7) {
        if constexpr (DIST == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      dma(t + DIST);
    }
    f16v st[2];
    i4 kfr[2][4], vfr[2][2][2];
    if constexpr (VARIANT == 5) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s = 0; s < 4; ++s) kfr[sub][s] = *(const i4*)(tile_img + (koffp[sub] ^ (s << 5)));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        i4 a = kf[sub];
        if constexpr (VARIANT == 4 || STREAM) a = *(const i4*)(tile_img + (STREAM ? (t % NST) * 16384 : 0) + (koffp[sub] ^ (s << 5)));
        if constexpr (VARIANT == 5) a = kfr[sub][s];
        st[sub] = mma(a, qf[s], (M16 ? s < 2 : s == 0) ? (s == 0 ? (f16v){0.f} : st[sub]) : st[sub], s & 1, &qf[(s + 2) & 3]);
      }
    }
    if constexpr (VARIANT == 5) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vfr[sub][s2][dt] = *(const i4*)(tile_img + (voffp[dt] ^ ((sub * 4 + 2 * s2) << 4)));
      __builtin_amdgcn_sched_barrier(0);
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      i4 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float sv = st[sub][8 * s2 + j];
          p[j] = (VARIANT == 1 || VARIANT == 2) ? sv : __builtin_amdgcn_exp2f(sv);  // variants 0, 3, 4, 5 exponentiate
          if (VARIANT != 1 && VARIANT != 3 && !DOT2) ps[j & 3] += p[j];
        }
        if (VARIANT != 1) {
          pf[s2][0] = pack2(p[0], p[1]);
          pf[s2][1] = pack2(p[2], p[3]);
          pf[s2][2] = pack2(p[4], p[5]);
          pf[s2][3] = pack2(p[6], p[7]);
          if (DOT2 && VARIANT != 3) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            const bf2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
            for (int w = 0; w < 4; ++w) ps[w] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pf[s2][w]), one, ps[w], false);
          }
        } else {  // MFMAs only: the accumulator's raw bits are the next B operand (keeps the S -> P.V dependency)
          pf[s2][0] = __builtin_bit_cast(int, p[0]);
          pf[s2][1] = __builtin_bit_cast(int, p[2]);
          pf[s2][2] = __builtin_bit_cast(int, p[4]);
          pf[s2][3] = __builtin_bit_cast(int, p[6]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          i4 a = vf[dt];
          if constexpr (VARIANT == 4 || STREAM) a = *(const i4*)(tile_img + (STREAM ? (t % NST) * 16384 : 0) + (voffp[dt] ^ ((sub * 4 + 2 * s2) << 4)));
          if constexpr (VARIANT == 5) a = vfr[sub][s2][dt];
          o[dt] = mma(a, pf[s2], o[dt], s2, &pf[s2 ^ 1]);
        }
    }
    l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    // keep the next tile's scores finite and data-dependent without extra vector work: rotate the q fragments (scalar-free moves
    // are folded by the compiler into operand choice)
    const i4 q0 = qf[0];
    qf[0] = qf[1]; qf[1] = qf[2]; qf[2] = qf[3]; qf[3] = q0;
  }
  float acc = l_run;
  for (int r = 0; r < 16; ++r) acc += o[0][r] + o[1][r];
  if (acc == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = acc;  // never true: keeps everything alive
}

static const char* g_big = nullptr;

template <int VARIANT, int WPS>
static double run(const int* seed, float* out, int tiles) {
  // WPS waves per SIMD = WPS workgroups of 4 waves per CU; 256 CUs
  const int blocks = 256 * WPS;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((mix_kernel<VARIANT, WPS>), dim3(blocks), dim3(256), 0, 0, seed, out, tiles, g_big);  // warm-up
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((mix_kernel<VARIANT, WPS>), dim3(blocks), dim3(256), 0, 0, seed, out, tiles, g_big);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double flops = 16.0 * 2.0 * 32 * 32 * 16 * (double)tiles * 4.0 * blocks;  // 16 MFMAs per wave and tile
  return flops / (best * 1e-3) / 1e12;
}

int main() {
  int* seed;
  float* out;
  hipMalloc(&seed, 1 << 20);
  char* big;
  hipMalloc(&big, (size_t)1 << 29);
  g_big = big;
  hipMalloc(&out, 256 * 4 * 256 * 4);
  // RANDOM bf16 operands (sign and mantissa random, magnitude 2^-7 .. 2^-5: scores stay small): the clock the chip holds under MFMA
  // load depends on operand toggling (MI355X_MICROARCH.md, DVFS give-back) -- constant data would flatter every row
  auto fill = [](void* dev, size_t bytes) {
    unsigned short* h = (unsigned short*)malloc(bytes);
    unsigned x = 12345u;
    for (size_t i = 0; i < bytes / 2; ++i) {
      x = x * 1664525u + 1013904223u;
      const unsigned r = x >> 8;
      h[i] = (unsigned short)(((r & 1) << 15) | ((0x78 + (r >> 1) % 3) << 7) | ((r >> 4) & 0x7f));
    }
    hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice);
    free(h);
  };
  fill(seed, 1 << 20);
  fill(big, (size_t)1 << 29);
  const int tiles = 2000;
  const char* names[12] = {"full mix (16 MFMA + 32 exp + 35 add + 16 pack)", "MFMAs only", "no exponentials", "no row sums",
                          "full mix + 16 ds_read_b128, read before use", "full mix + 16 ds_read_b128, whole tile in flight",
                          "... + 4 LDS-DMA per wave and tile, issue only", "... + 4 LDS-DMA + vmcnt(0) + barrier per tile (L2-resident source)",
                          "... the same, strided K / V^T rows, L2-resident after the first pass", "... the same, contiguous tiles streamed from HBM by EVERY workgroup (HBM-bound)",
                          "... the same, 5 workgroups share a (sequence, head): tiles new to L2, two stages (= the kernel)", "... the same with three stages (tile t+2 requested at tile t)"};
  printf("attention structure probe, random bf16 operands, 256 CUs, TFLOP/s of the 16 MFMAs per tile (nominal peak 2500):\n");
#define ROW(V)                                                                                                              \
  printf("  %-98s 1 wave/SIMD %7.0f   2 %7.0f   3 %7.0f   4 %7.0f\n", names[V], run<V, 1>(seed, out, tiles), run<V, 2>(seed, out, tiles), \
         run<V, 3>(seed, out, tiles), run<V, 4>(seed, out, tiles));
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11)
  return 0;
}
