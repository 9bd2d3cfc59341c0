// K5: fused multi-head attention for gfx950, bf16 MFMA operands / fp32 softmax and accumulation.
//
//   out[q, h*64 + d] = sum_k softmax_k(q.k / 8) * v[k, d]        head_dim = 64 (ViT-L/16, ViT-S/14)
//
// Layout contract (produced by the QKV GEMM epilogue, gemm.hip EPI_QKV):
//   qk  [rows, 2D]  bf16, row = seq*S + token; q at column h*64, k at column D + h*64
//   vT  [seq][head][64][kpad] bf16: V transposed so that keys are contiguous (the P.V MFMA wants
//       both operands K-contiguous; the transpose is paid once in the GEMM epilogue).
//
// One workgroup = 4 waves = 128 consecutive queries of one (sequence, head); a wave owns 32
// queries. K and V^T tiles of 64 keys are streamed through a 2-stage LDS ring with
// global_load_lds (swizzled on the source address like the GEMM), shared by the 4 waves.
// "Swapped" QK^T: S^T = K.Q^T puts one query per lane (column) and 16 keys per lane in the
// accumulator registers, so the online-softmax max/sum are in-register reductions plus ONE
// cross-half exchange, and the exponentiated accumulator registers are -- after a bf16 pack --
// directly the B operand of the P.V MFMA (no LDS round trip, no cross-lane movement). The key
// order inside a 32-key sub-tile is permuted (bits 2 and 3 of the key index swapped when K rows
// are read from LDS) so that registers 8s..8s+7 hold the 8 consecutive keys the V^T fragment
// of k-step s holds.
#include "ops.h"
#include "elem.h"

namespace md {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// v_mfma_f32_32x32x16 on bf16 or IEEE half fragments (same rate, same layout)
template <typename T>
__device__ __forceinline__ f32x16_t mfma32(const i32x4_t& a, const i32x4_t& b, const f32x16_t& c) {
  typedef typename Native<T>::v8 v8;
  if constexpr (std::is_same<T, bf16_t>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void glds16a(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// grid: 1-D, q blocks of 128 x heads x sequences; block 256. The block -> (unit, q block) map is
// XCD-aware: blocks b and b+8 share an XCD, so logical ids are dealt in contiguous ranges per XCD and
// the q blocks of one (sequence, head) run on ONE XCD -- its K and V^T (148 KB) are fetched into that
// L2 once instead of once per q block (measured before the remap: 1.0 GB fetched per launch against
// 0.27 GB of q/k/v).
template <typename T, bool FP8OUT>
__global__ __launch_bounds__(256, 2) void attention_kernel(const T* __restrict__ qk,
                                                             const T* __restrict__ vT, T* __restrict__ out,
                                                             int S, int n_tokens, int heads, int D, int kpad,
                                                             int qblocks, float out_fp8_inv) {
  constexpr int STAGE = 16384;  // K tile 64x128B + V^T tile 64x128B
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int unit = id / qblocks, qb = id - unit * qblocks;
  const int seq = unit / heads, head = unit - seq * heads;
  const int q0 = qb * 128 + wave * 32;
  const bool active = q0 < n_tokens;  // wave-uniform
  const int h = lane >> 5, c = lane & 31;
  const long two_d = 2L * D;
  const long seq_row0 = (long)seq * S;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane (c,h) holds Q[q0+c][16s + 8h + j] ----
  i32x4_t qf[4];
  {
    int q = q0 + c;
    q = q < n_tokens ? q : n_tokens - 1;
    const char* qp = (const char*)(qk + (seq_row0 + q) * two_d + head * 64) + h * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const i32x4_t*)(qp + s * 32);
  }

  // ---- global->LDS sources: 16 row-groups per stage (8 K + 8 V^T), 4 per wave ----
  const int lrow = lane >> 3, pc = lane & 7;
  const char* ksrc[2];
  const char* vsrc[2];
  int kmaxrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (i * 4 + wave) * 8 + lrow;  // tile row 0..63
    const int lc = pc ^ ((r >> 1) & 7);
    ksrc[i] = (const char*)(qk + D + head * 64) + lc * 16;  // + key row * two_d * 2 per tile
    kmaxrow[i] = r;
    vsrc[i] = (const char*)(vT + (((long)seq * heads + head) * 64 + r) * kpad) + lc * 16;  // + kv0*2 per tile
  }

  auto issue = [&](int stage, int t) {
    char* sb = smem + stage * STAGE;
    const int kv0 = t * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kv0 + kmaxrow[i];
      key = key < n_tokens ? key : n_tokens - 1;  // clamped rows are masked below
      glds16a(ksrc[i] + (seq_row0 + key) * two_d * 2, sb + (i * 4 + wave) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16a(vsrc[i] + (long)kv0 * 2, sb + 8192 + (i * 4 + wave) * 1024);
  };

  // LDS read offsets. K rows are read through the bit-2/bit-3 swap; V^T rows (= d) directly.
  const int pk = (c & 0x13) | ((c & 4) << 1) | ((c & 8) >> 1);
  int koff[2], voff[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int R = sub * 32 + pk;
    koff[sub] = R * 128 + ((((R >> 1) & 7) ^ h) << 4);  // chunk (2s+h) ^ swz -> xor (s<<5) per k-step
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int R = dt * 32 + c;
    voff[dt] = 8192 + R * 128 + ((((R >> 1) & 7) ^ h) << 4);  // chunk (sub*4 + 2s' + h) ^ swz
  }

  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float cs = 0.125f * 1.4426950408889634f;  // head_dim^-0.5 * log2(e)

  const int NT = (n_tokens + 63) / 64;
  issue(0, 0);
  for (int t = 0; t < NT; ++t) {
    const int cur = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < NT) issue(cur ^ 1, t + 1);
    if (!active) continue;
    const char* sb = smem + cur * STAGE;

    // ---- S^T[sub] = K[sub] . Q^T ----
    f32x16_t st[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const i32x4_t kf = *(const i32x4_t*)(sb + (koff[sub] ^ (s << 5)));
        const f32x16_t cin = s == 0 ? (f32x16_t){0.f} : st[sub];  // first k-step: inline-constant 0 as C
        st[sub] = mfma32<T>(kf, qf[s], cin);
      }
    }
    // register r of lane half h holds local key (r&7) + 8h + 16(r>>3) of the sub-tile
    const int kv0 = t * 64;
    if (kv0 + 64 > n_tokens) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + sub * 32 + (r & 7) + 8 * h + 16 * (r >> 3);
          if (key >= n_tokens) st[sub][r] = -INFINITY;
        }
    }
    // ---- online softmax (one query per lane; the two lane halves share a query) ----
    // Deferred rescale: the running max m_run is only raised (and O, l rescaled) when some row of the
    // wave saw a score more than 2^kDefer above it; otherwise exponentials stay relative to the old
    // max (p <= 2^kDefer, harmless for bf16's floating-point rounding and the fp32 sums).  With
    // softmax logits of O(1) spread this fires on the first tile and then almost never, which removes
    // the 32-register O rescale from nearly every tile.  The final result is exact either way.
    constexpr float kDefer = 6.0f;  // in log2 units
    float mx = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
#pragma unroll
    for (int r = 3; r + 1 < 16; r += 2) mx = fmaxf(fmaxf(mx, st[0][r]), st[0][r + 1]);
    mx = fmaxf(mx, st[0][15]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[1][r]), st[1][r + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any((mx - m_run) * cs > kDefer)) {  // wave-uniform branch
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float mc = m_run * cs;
    float psum = 0.f;
    i32x4_t pf[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(st[sub][r] * cs - mc);
        psum += p[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        pf[sub][s2][0] = pack2_nosat<T>(p[8 * s2 + 0], p[8 * s2 + 1]);
        pf[sub][s2][1] = pack2_nosat<T>(p[8 * s2 + 2], p[8 * s2 + 3]);
        pf[sub][s2][2] = pack2_nosat<T>(p[8 * s2 + 4], p[8 * s2 + 5]);
        pf[sub][s2][3] = pack2_nosat<T>(p[8 * s2 + 6], p[8 * s2 + 7]);
      }
    }
    l_run += psum;
    // ---- O^T[dt] += V^T[dt] . P^T ----
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const i32x4_t vf = *(const i32x4_t*)(sb + (voff[dt] ^ ((sub * 4 + 2 * s2) << 4)));
          o[dt] = mfma32<T>(vf, pf[sub][s2], o[dt]);
        }
  }

  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv_l = 1.0f / l_tot;
  const int q = q0 + c;
  if constexpr (FP8OUT) {  // e4m3 rows for an fp8-operand output projection
    if (q < n_tokens) {
    char* orow8 = (char*)out + (seq_row0 + q) * (long)D + head * 64;
    const float sc = inv_l * out_fp8_inv;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * h;
        auto cl = [&](float a) { return __builtin_amdgcn_fmed3f(a * sc, -448.f, 448.f); };
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(o[dt][4 * q4]), cl(o[dt][4 * q4 + 1]), 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(o[dt][4 * q4 + 2]), cl(o[dt][4 * q4 + 3]), w, true);
        *(int*)(orow8 + d) = w;
      }
    }
  } else if (q < n_tokens) {
    T* orow = out + (seq_row0 + q) * (long)D + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * h;
        store4<T>(orow + d, (f32x4_t){o[dt][4 * q4] * inv_l, o[dt][4 * q4 + 1] * inv_l, o[dt][4 * q4 + 2] * inv_l, o[dt][4 * q4 + 3] * inv_l});
      }
  }
}

int launch_attention(const void* qk, const void* vT, void* out, int nseq, int S, int n_tokens, int heads, int D,
                     int kpad, int prec, hipStream_t s, float out_fp8_inv) {
  if (prec != MD_PREC_BF16 && prec != MD_PREC_F16) MD_FAIL(MD_ERR_UNSUPPORTED, "fused attention takes bf16 or f16 operands (precision %d)", prec);
  if (D != heads * 64) MD_FAIL(MD_ERR_UNSUPPORTED, "attention: head_dim must be 64 (D=%d heads=%d)", D, heads);
  if (kpad % 64 != 0 || kpad < (n_tokens + 63) / 64 * 64)
    MD_FAIL(MD_ERR_INVALID_ARG, "attention: kpad=%d must be a multiple of 64 covering %d keys", kpad, n_tokens);
  const int qblocks = (n_tokens + 127) / 128;
  const long blocks = (long)qblocks * heads * nseq;
  if (nseq <= 0 || blocks > 0x7fffffffL) MD_FAIL(MD_ERR_UNSUPPORTED, "attention: %d sequences", nseq);
  if (out_fp8_inv > 0.f) {
    if (prec != MD_PREC_BF16) MD_FAIL(MD_ERR_UNSUPPORTED, "attention: e4m3 output rows are built for bf16 operands");
    hipLaunchKernelGGL((attention_kernel<bf16_t, true>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)qk, (const bf16_t*)vT,
                       (bf16_t*)out, S, n_tokens, heads, D, kpad, qblocks, out_fp8_inv);
  } else if (prec == MD_PREC_F16) {
    hipLaunchKernelGGL((attention_kernel<f16_t, false>), dim3((unsigned)blocks), dim3(256), 0, s, (const f16_t*)qk, (const f16_t*)vT,
                       (f16_t*)out, S, n_tokens, heads, D, kpad, qblocks, out_fp8_inv);
  } else {
    hipLaunchKernelGGL((attention_kernel<bf16_t, false>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)qk, (const bf16_t*)vT,
                       (bf16_t*)out, S, n_tokens, heads, D, kpad, qblocks, out_fp8_inv);
  }
  MD_HIP(hipGetLastError());
  return MD_OK;
}

}  // namespace md
