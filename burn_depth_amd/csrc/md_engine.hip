// Depth Pro engine for MI355X: weights resident in HBM (fp32 master + MFMA-operand copies),
// a static workspace plan sized for max_batch, and the forward schedule of DepthPro::infer
// (depth_pro/mod.rs:312-364) expressed as launches of the hand-written gfx950 kernels.
//
// Data layout in HBM:
//   * ViT token tensors are row-major [sequence*SS + token, channels]; SS = tokens rounded up to 4
//     (580 for 577).  The three ViT-L encoders (patch / image / fov; encoder.rs:346-348,409,
//     fov.rs:203) are "row groups" of the same tensors and advance through ONE launch per op.
//   * the residual stream is fp32; every GEMM operand is T (bf16 or f32 by precision mode).
//   * feature maps are NHWC (pixel-major rows of channels): a 1x1 conv is a dense GEMM, a k2s2
//     ConvTranspose a GEMM + pixel-shuffle epilogue, a 3x3 conv an implicit GEMM over taps, and
//     the token->map reshape + overlap-trim merge (encoder.rs:234-319) a row-index table.
#include "md_engine.h"
#include "md_engine_util.h"
#include "kernels/elem.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace md {

// ------------------------------------------------------------------------------------------------
// geometry helpers
// ------------------------------------------------------------------------------------------------
void split_geometry(int image_size, int patch_size, float overlap, int* stride, int* steps) {
  int st = (int)std::floor((float)patch_size * (1.0f - overlap));  // encoder.rs:196-197
  st = std::max(st, 1);
  st = std::min(st, patch_size);
  *stride = st;
  *steps = patch_size >= image_size ? 1 : 1 + (image_size - patch_size + st - 1) / st;
}

int feature_padding(int patch_size, int stride, int fps) {  // encoder.rs:28-38
  if (fps == 0 || patch_size == 0) return 0;
  const int denom = std::max(patch_size, 1);
  const int fstride = (stride * fps + denom / 2) / denom;
  return std::max(fps - fstride, 0) / 2;
}

int merged_extent(int h, int steps, int pad) { return steps > 1 ? steps * (h - 2 * pad) + 2 * pad : h; }

void merge_source(int Y, int X, int h, int w, int steps, int pad, int* j, int* i, int* ty, int* tx) {
  int jj = 0, ii = 0;
  if (steps > 1) {
    const int ih = h - 2 * pad, iw = w - 2 * pad;
    jj = Y < pad ? 0 : std::min((Y - pad) / ih, steps - 1);
    ii = X < pad ? 0 : std::min((X - pad) / iw, steps - 1);
    *ty = Y - jj * ih;
    *tx = X - ii * iw;
  } else {
    *ty = Y;
    *tx = X;
  }
  *j = jj;
  *i = ii;
}

// ------------------------------------------------------------------------------------------------
// weight packing kernels: fp32 master -> MFMA operand layouts
// ------------------------------------------------------------------------------------------------
// `terms` copies of the contraction per output row (MD_PREC_F16X2): blocks 0 and 1 hold f16(w) -- they multiply the hi and
// the lo plane of the activation row --, block 2 holds f16(w - f16(w)) and multiplies the hi plane again (GemmParams::a_wrap).
template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int kind, int d0, int d1, int k,
                            int kp, int terms, long total) {
  const int kpt = kp * terms;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    int blk = 0;
    if (kind == PACK_NK) {  // [N][K] -> [N][terms][kp]
      const int kq = (int)(e % kpt);
      blk = kq / kp;
      const int kk = kq - blk * kp;
      const long n = e / kpt;
      if (kk < d1) v = src[n * d1 + kk];
    } else if (kind == PACK_CONV3) {  // [Cout][Cin][3][3] -> [Cout][9][terms][kp]
      const int cq = (int)(e % kpt);
      blk = cq / kp;
      const int ci = cq - blk * kp;
      const long t = e / kpt;
      const int tap = (int)(t % 9);
      const long co = t / 9;
      if (ci < d1) v = src[(co * d1 + ci) * 9 + tap];
    } else if (kind == PACK_DECONV) {  // [Cin][Cout][k][k] -> [k*k*Cout][terms][kp], row = tap*Cout + co
      const int cq = (int)(e % kpt);
      blk = cq / kp;
      const int ci = cq - blk * kp;
      const long n = e / kpt;
      const int tap = (int)(n / d1);
      const int co = (int)(n % d1);
      if (ci < d0) v = src[((long)ci * d1 + co) * (k * k) + tap];
    } else {  // PACK_DIRECT: [Cout][Cin][k][k] -> [Cout][k][k][Cin]
      const int ci = (int)(e % d1);
      long t = e / d1;
      const int kx = (int)(t % k);
      t /= k;
      const int ky = (int)(t % k);
      const long co = t / k;
      v = src[((co * d1 + ci) * k + ky) * k + kx];
    }
    if constexpr (is_split<T>::value) {
      const _Float16 h = cvt_elem<T>(v);
      *(_Float16*)(dst + e) = blk < 2 ? h : cvt_elem<T>(v - (float)h);
    } else {
      st1<T>(dst + e, v);
    }
  }
}

// MD_PREC_F16X2: number of values of w[0..n) that are not exactly representable as an IEEE half (the reference's
// checkpoints are f16 records, mod.rs:206: none; an fp32 checkpoint or the seeded test weights: nearly all)
__global__ void count_inexact_f16_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ count) {
  unsigned c = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = w[i];
    c += ((float)(_Float16)v != v) ? 1u : 0u;
  }
  if (c) atomicAdd(count, c);
}
// w[i] = f32(f16(w[i])): what reading the value back from an f16 record gives
__global__ void round_f16_kernel(float* __restrict__ w, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    w[i] = (float)(_Float16)__builtin_amdgcn_fmed3f(w[i], -65504.f, 65504.f);
}

// W'[ci][co][q] = sum_m Wd[ci][m][q] * Wo[co][m]; Wd is [Cin, Cmid, 2, 2], Wo is [Cout, Cmid] (1x1 conv)
__global__ void compose_deconv_conv_kernel(const float* __restrict__ wd, const float* __restrict__ wo, int cin, int cmid,
                                           int cout, float* __restrict__ out) {
  const long total = (long)cin * cout * 4;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e & 3);
    const int co = (int)((e >> 2) % cout);
    const int ci = (int)((e >> 2) / cout);
    float acc = 0.f;
    for (int mth = 0; mth < cmid; ++mth) acc += wd[((long)ci * cmid + mth) * 4 + q] * wo[(long)co * cmid + mth];
    out[e] = acc;
  }
}

// Wa [Cin, Cmid, 2, 2], Wb [Cmid, Cout, 2, 2] (both bias-free ConvTranspose2d k2 s2) -> out [Cin, Cout, 4, 4]:
// output pixel (4y + 2 dy1 + dy2, 4x + 2 dx1 + dx2) = sum_m (in[y, x] . Wa[:, m, dy1, dx1]) * Wb[m, co, dy2, dx2]
__global__ void compose_deconv_pair_kernel(const float* __restrict__ wa, const float* __restrict__ wb, int cin, int cmid, int cout,
                                           float* __restrict__ out) {
  const long total = (long)cin * cout * 16;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tx = (int)(e & 3), ty = (int)((e >> 2) & 3);
    const int co = (int)((e >> 4) % cout);
    const int ci = (int)((e >> 4) / cout);
    const int qa = (ty >> 1) * 2 + (tx >> 1), qb = (ty & 1) * 2 + (tx & 1);
    float acc = 0.f;
    for (int mth = 0; mth < cmid; ++mth) acc += wa[((long)ci * cmid + mth) * 4 + qa] * wb[((long)mth * cout + co) * 4 + qb];
    out[e] = acc;
  }
}

int compose_deconv_pair(const float* wa, const float* wb, int cin, int cmid, int cout, float* out, hipStream_t s) {
  const long total = (long)cin * cout * 16;
  hipLaunchKernelGGL(compose_deconv_pair_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, wa, wb, cin,
                     cmid, cout, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int compose_deconv_conv(const float* wd, const float* wo, int cin, int cmid, int cout, float* out, hipStream_t s) {
  const long total = (long)cin * cout * 4;
  hipLaunchKernelGGL(compose_deconv_conv_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, wd,
                     wo, cin, cmid, cout, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// Depth head, mod.rs:105-108: ConvTranspose2d k2 s2 (Wd [Cin,Cmid,2,2], bias bd) -> Conv2d 3x3 pad 1 (W1 [Cout,Cmid,3,3],
// bias b1) with nothing between them. Output pixel (2y+py, 2x+px) of the pair reads deconv pixels (2y+py+u-1, 2x+px+v-1),
// u, v in 0..2, i.e. input pixels (y+a, x+b) with a = floor((py+u-1)/2), through deconv tap ((py+u-1)&1, (px+v-1)&1):
// the pair is ONE 3x3 convolution on the deconv's INPUT grid with 4*Cout output columns, column (py, px, co):
//   Wc[(2py+px)*Cout + co][ci][a+1][b+1] = sum_{u -> a} sum_{v -> b} sum_mid W1[co][mid][u][v] * Wd[ci][mid][dy][dx]
// (4 of the 9 input taps are non-zero per parity). `out` has the [N][Cin][3][3] layout pack_kernel(PACK_CONV3) reads.
__global__ void compose_head_kernel(const float* __restrict__ wd, const float* __restrict__ w1, int cin, int cmid, int cout,
                                    float* __restrict__ out) {
  const long total = 4L * cout * cin * 9;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % 9);
    const int ci = (int)((e / 9) % cin);
    const int n = (int)(e / 9 / cin);
    const int q = n / cout, co = n - q * cout, py = q >> 1, px = q & 1;
    const int a = tap / 3 - 1, b = tap % 3 - 1;
    float acc = 0.f;
    for (int u = 0; u < 3; ++u) {
      if ((py + u + 1) / 2 - 1 != a) continue;
      const int dy = (py + u + 1) & 1;
      for (int v = 0; v < 3; ++v) {
        if ((px + v + 1) / 2 - 1 != b) continue;
        const int dx = (px + v + 1) & 1;
        for (int mid = 0; mid < cmid; ++mid)
          acc += w1[(((long)co * cmid + mid) * 3 + u) * 3 + v] * wd[(((long)ci * cmid + mid) * 2 + dy) * 2 + dx];
      }
    }
    out[e] = acc;
  }
}

// The deconv bias reaches the conv through every tap that lies INSIDE the 2H x 2W map (the conv zero-pads the deconv's
// output, bias included), so the pair's bias depends on the position class of the output pixel:
//   bias[3*ry + rx][co] = b1[co] + sum_{u valid for ry} sum_{v valid for rx} sum_mid W1[co][mid][u][v] * bd[mid]
// ry = 0 first row (u = 0 outside), 1 interior, 2 last row (u = 2 outside); rx likewise.
__global__ void compose_head_bias_kernel(const float* __restrict__ w1, const float* __restrict__ bd, const float* __restrict__ b1,
                                         int cmid, int cout, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 9 * cout) return;
  const int cls = e / cout, co = e - cls * cout, ry = cls / 3, rx = cls % 3;
  float acc = b1[co];
  for (int u = 0; u < 3; ++u) {
    if ((ry == 0 && u == 0) || (ry == 2 && u == 2)) continue;
    for (int v = 0; v < 3; ++v) {
      if ((rx == 0 && v == 0) || (rx == 2 && v == 2)) continue;
      for (int mid = 0; mid < cmid; ++mid) acc += w1[(((long)co * cmid + mid) * 3 + u) * 3 + v] * bd[mid];
    }
  }
  out[e] = acc;
}

// conv 1x1 (W1 [Cmid,Cin], bias) -> conv 3x3 (W3 [Cout,Cmid,3,3]): Wc[co][ci][tap] = sum_m W3[co][m][tap] * W1[m][ci];
// the bias classes are compose_head_bias_kernel's formula with (W3, the 1x1's bias, the 3x3's bias).
__global__ void compose_c1c3_kernel(const float* __restrict__ w1, const float* __restrict__ w3, int cin, int cmid, int cout,
                                    float* __restrict__ out) {
  const long total = (long)cout * cin * 9;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % 9);
    const int ci = (int)((e / 9) % cin);
    const int co = (int)(e / 9 / cin);
    float acc = 0.f;
    for (int mth = 0; mth < cmid; ++mth) acc += w3[((long)co * cmid + mth) * 9 + tap] * w1[(long)mth * cin + ci];
    out[e] = acc;
  }
}

int compose_c1c3(const float* w1, const float* w3, int cin, int cmid, int cout, float* out, hipStream_t s) {
  const long total = (long)cout * cin * 9;
  hipLaunchKernelGGL(compose_c1c3_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, w1, w3, cin, cmid,
                     cout, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int compose_head(const float* wd, const float* w1, int cin, int cmid, int cout, float* out, hipStream_t s) {
  const long total = 4L * cout * cin * 9;
  hipLaunchKernelGGL(compose_head_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, wd, w1, cin,
                     cmid, cout, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int compose_head_bias(const float* w1, const float* bd, const float* b1, int cmid, int cout, float* out, hipStream_t s) {
  hipLaunchKernelGGL(compose_head_bias_kernel, dim3((9 * cout + 255) / 256), dim3(256), 0, s, w1, bd, b1, cmid, cout, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int count_inexact_f16(const float* w, long n, hipStream_t s, unsigned* out) {
  unsigned* d_cnt = nullptr;
  MD_HIP(hipMalloc((void**)&d_cnt, 4));
  MD_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
  hipLaunchKernelGGL(count_inexact_f16_kernel, dim3((int)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, s, w, n, d_cnt);
  MD_HIP(hipMemcpyAsync(out, d_cnt, 4, hipMemcpyDeviceToHost, s));
  MD_HIP(hipStreamSynchronize(s));
  MD_HIP(hipFree(d_cnt));
  return MD_OK;
}

int pack_weight(const float* src, const PackEntry& e, int prec, hipStream_t s) {
  const int pk_prec = e.f32 ? MD_PREC_F32 : prec;
  const int terms = pk_prec == MD_PREC_F16X2 ? e.terms : 1;
  if (pk_prec == MD_PREC_F16X2 && terms != 2 && terms != 3) MD_FAIL(MD_ERR_INVALID_ARG, "pack: split-half weights take 2 or 3 terms, got %d", terms);
  const long total = (long)pack_elems(e, terms);
  const int grid = (int)std::min<long>((total + 255) / 256, 4096);
  MD_BY_PREC(pk_prec, hipLaunchKernelGGL(pack_kernel<T>, dim3(grid), dim3(256), 0, s, src, (T*)e.dst, e.kind, e.d0, e.d1, e.k, e.kp, terms, total));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

}  // namespace md

using namespace md;

// ------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------
struct md_model_s::Buffers {
  float* xin = nullptr;       // [B,3,S,S] fp32 (resized / staged input)
  float* xraw = nullptr;      // staged raw input when it arrives on the host or needs resizing
  uint8_t* rgb = nullptr;     // staged RGB bytes
  void* patches = nullptr;    // [nseq_p*P, Kpe] T
  float* xres = nullptr;      // [nseq*SS, D] fp32 residual stream
  void* xn = nullptr;         // [nseq*SS, D] T
  float* ln_stats = nullptr;  // [nseq*SS, D/256, 2] fp32: per row and 256-column tile (mean, centred sum of squares) of the residual stream (LayerNorm fold)
  float* ln_ab = nullptr;     // [nseq*SS, 2] fp32: (rstd, -mu rstd) per row, finished from ln_stats between the producer and the consumer GEMM
  void* qk = nullptr;         // [nseq*SS, 2D] T
  void* vT = nullptr;         // [nseq][heads][64][kpad] T
  void* ao = nullptr;         // [nseq*SS, D] T
  int* attn_redo = nullptr;   // [attention_redo_ints(nseq*heads)] flags + compacted list of the assembly attention kernel (raised = that unit re-runs in the safe body)
  void* hbuf = nullptr;       // [nseq*SS, 4D] T
  float* scores = nullptr;    // fp32 attention only
  void* hook[2] = {nullptr, nullptr};  // [n0*SS, D] T
  void* tok = nullptr;        // [nseq*SS, D] T (final-norm tokens)
  // encoder
  void *l0p = nullptr, *l0a = nullptr, *l0b = nullptr, *enc0 = nullptr, *enc0r = nullptr;
  void *l1p = nullptr, *l1a = nullptr, *enc1 = nullptr;
  void *x0p = nullptr, *enc2 = nullptr;
  void *x1p = nullptr, *enc3 = nullptr;
  void *x2p = nullptr, *cat = nullptr, *enc4 = nullptr;
  // decoder (per level l: proj/projr, t, x/xr, t2, y, up, f)
  void *proj[5] = {0, 0, 0, 0, 0}, *projr[5] = {0, 0, 0, 0, 0};
  void *dt[5] = {0, 0, 0, 0, 0}, *dx[5] = {0, 0, 0, 0, 0}, *dxr[5] = {0, 0, 0, 0, 0};
  void *dy[5] = {0, 0, 0, 0, 0}, *dup[5] = {0, 0, 0, 0, 0}, *df[5] = {0, 0, 0, 0, 0};
  // head
  void *h0 = nullptr, *h1 = nullptr;
  float* canonical = nullptr;  // [B, S*S]
  float* inv = nullptr;        // [B, S*S] (resize path)
  float* depth_stage = nullptr;  // device staging for host outputs [B, Hmax*Wmax]
  // grow-only capacities (bytes) of the staging buffers that serve host pointers and non-native input sizes, and the pinned
  // bounce buffers between pageable caller memory and the DMA engine: a steady stream of same-sized calls allocates nothing
  size_t rgb_cap = 0, xraw_cap = 0;
  void *pin_in = nullptr, *pin_out = nullptr;
  size_t pin_in_cap = 0, pin_out_cap = 0;
  // fov
  float *fovproj = nullptr, *fv0 = nullptr, *fv1 = nullptr, *fv2 = nullptr, *fv3 = nullptr, *fvr = nullptr;
  float *fov_deg = nullptr, *focal = nullptr, *fovy = nullptr, *ratio = nullptr;
  size_t depth_stage_elems = 0;
};

namespace md {


static void level_sizes(const md_model_s* m, int lvl_hw[5]) {
  // spatial size of encoder feature l (encoder.rs:416-434): latent0 x8, latent1 x4, x0 x2, x1 x2(mid), fused x2
  lvl_hw[0] = m->mh_hi * 8;
  lvl_hw[1] = m->mh_hi * 4;
  lvl_hw[2] = m->mh_hi * 2;
  lvl_hw[3] = m->mh_mid * 2;
  lvl_hw[4] = m->g * 2;
}

static int plan_workspace(md_model_s* m, bool dry, size_t* total_out) {
  const ModelCfg& c = m->cfg;
  const int B = c.max_batch, D = c.pv.D, F = c.F, P = m->P, SS = m->SS;
  const int esz = m->esz * m->xm;  // bytes per LOGICAL element of a T tensor (MD_PREC_F16X2: two half planes)
  const int n0 = m->steps0 * m->steps0 * B, n1 = m->steps1 * m->steps1 * B;
  const int nseq_p = n0 + n1 + B;
  const int nseq = nseq_p + B * (m->ngroups - 1);
  const int* dims = c.pv.feat_dims;
  auto cp = [&](int ch) { return round_up(ch, m->ke); };
  md_model_s::Buffers* b = m->buf;
  size_t total = 0;
  auto take = [&](size_t bytes) -> void* {
    bytes = align_up(bytes + 256, 256);  // slack: clamped tile rows never cross an allocation
    total += bytes;
    if (dry) return nullptr;
    return m->ws.take(bytes);
  };
#define MD_TAKE(field, type, bytes)                                         \
  do {                                                                      \
    void* _p = take(bytes);                                                 \
    if (!dry) {                                                             \
      if (!_p) MD_FAIL(MD_ERR_OOM, "workspace arena exhausted at " #field); \
      b->field = (type)_p;                                                  \
    }                                                                       \
  } while (0)
  const size_t S2 = (size_t)m->S * m->S;
  const int Kpe = 3 * c.pv.ps * c.pv.ps;
  MD_TAKE(xin, float*, (size_t)B * 3 * S2 * 4);
  MD_TAKE(patches, void*, (size_t)nseq_p * P * Kpe * esz);
  const size_t rows = (size_t)nseq * SS + 64;
  MD_TAKE(xres, float*, rows * D * 4);
  MD_TAKE(xn, void*, rows * D * esz);
  if (m->ln_fold_can) MD_TAKE(ln_stats, float*, rows * (D / 256) * 8);
  if (m->ln_fold_can) MD_TAKE(ln_ab, float*, rows * 8);
  MD_TAKE(qk, void*, rows * 2 * D * esz);
  MD_TAKE(vT, void*, (size_t)nseq * c.pv.heads * 64 * m->kpad * esz);
  m->vt_plane = m->xm == 2 ? (size_t)nseq * c.pv.heads * 64 * m->kpad : 0;
  MD_TAKE(ao, void*, rows * D * esz);
  MD_TAKE(attn_redo, int*, (size_t)attention_redo_ints(nseq * c.pv.heads) * 4);  // the assembly attention kernel's per-(sequence, head) flags (zero = the arena's memset)
  MD_TAKE(hbuf, void*, rows * 4 * D * esz);
  if (m->prec == MD_PREC_F32) MD_TAKE(scores, float*, (size_t)nseq * c.pv.heads * SS * m->kpad * 4);
  MD_TAKE(hook[0], void*, ((size_t)n0 * SS + 64) * D * esz);
  MD_TAKE(hook[1], void*, ((size_t)n0 * SS + 64) * D * esz);
  MD_TAKE(tok, void*, rows * D * esz);
  const size_t hi = (size_t)B * m->mh_hi * m->mh_hi, mid = (size_t)B * m->mh_mid * m->mh_mid,
               lo = (size_t)B * m->g * m->g;
  MD_TAKE(l0p, void*, hi * cp(dims[0]) * esz);
  MD_TAKE(l0a, void*, hi * 4 * cp(F) * esz);
  MD_TAKE(l0b, void*, hi * 16 * cp(F) * esz);
  MD_TAKE(enc0, void*, hi * 64 * cp(F) * esz);
  MD_TAKE(enc0r, void*, hi * 64 * cp(F) * esz);
  MD_TAKE(l1p, void*, hi * cp(dims[0]) * esz);
  MD_TAKE(l1a, void*, hi * 4 * cp(dims[0]) * esz);
  MD_TAKE(enc1, void*, hi * 16 * cp(dims[0]) * esz);
  MD_TAKE(x0p, void*, hi * cp(dims[1]) * esz);
  MD_TAKE(enc2, void*, hi * 4 * cp(dims[1]) * esz);
  MD_TAKE(x1p, void*, mid * cp(dims[2]) * esz);
  MD_TAKE(enc3, void*, mid * 4 * cp(dims[2]) * esz);
  MD_TAKE(x2p, void*, lo * cp(dims[3]) * esz);
  MD_TAKE(cat, void*, lo * 4 * 2 * cp(dims[3]) * esz);
  MD_TAKE(enc4, void*, lo * 4 * cp(dims[3]) * esz);
  int hw[5];
  level_sizes(m, hw);
  for (int l = 0; l < 5; ++l) {
    const size_t px = (size_t)B * hw[l] * hw[l];
    const size_t fb = px * cp(F) * esz;
    if (l != 0) {
      MD_TAKE(proj[l], void*, fb);
      MD_TAKE(projr[l], void*, fb);
    }
    MD_TAKE(dt[l], void*, fb);
    if (l != 4) {
      MD_TAKE(dx[l], void*, fb);
      MD_TAKE(dxr[l], void*, fb);
    }
    MD_TAKE(dy[l], void*, fb);
    if (l != 0) MD_TAKE(dup[l], void*, fb * 4);
    MD_TAKE(df[l], void*, (l != 0 ? fb * 4 : fb));
  }
  const size_t px0 = (size_t)B * hw[0] * hw[0];
  MD_TAKE(h0, void*, px0 * cp(F / 2) * esz);
  MD_TAKE(h1, void*, px0 * 4 * cp(F / 2) * esz);
  MD_TAKE(canonical, float*, px0 * 4 * 4);
  MD_TAKE(inv, float*, px0 * 4 * 4);
  if (c.use_fov_head) {
    const size_t l4 = (size_t)B * hw[4] * hw[4];
    MD_TAKE(fovproj, float*, (size_t)B * P * (F / 2) * 4);
    MD_TAKE(fv0, float*, l4 * F * 4);
    MD_TAKE(fv1, float*, l4 * F * 4);
    MD_TAKE(fv2, float*, l4 * F * 4);
    MD_TAKE(fv3, float*, l4 * F * 4);
    MD_TAKE(fvr, float*, (size_t)B * 36 * F * 4 + l4 * F * 4);
  }
  MD_TAKE(fov_deg, float*, (size_t)B * 4);
  MD_TAKE(focal, float*, (size_t)B * 4);
  MD_TAKE(fovy, float*, (size_t)B * 4);
  MD_TAKE(ratio, float*, (size_t)B * 4);
#undef MD_TAKE
  if (total_out) *total_out = total + 4096;
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// model construction
// ------------------------------------------------------------------------------------------------
int model_create(md_device_t dev, const ModelCfg& cfg, md_model_t* out) {
  if (!dev || !out) MD_FAIL(MD_ERR_INVALID_ARG, "device/model pointer is null");
  if (!cfg.iv.same_arch(cfg.pv) || (cfg.use_fov_head && cfg.has_fov_vit && !cfg.fv.same_arch(cfg.pv)))
    MD_FAIL(MD_ERR_UNSUPPORTED,
            "patch/image/fov encoders must share one ViT architecture (they run as row groups of one launch)");
  if (cfg.pv.D != cfg.pv.heads * 64) MD_FAIL(MD_ERR_UNSUPPORTED, "ViT head_dim must be 64");
  if (cfg.pv.D % 64 != 0 || cfg.pv.D > 1024) MD_FAIL(MD_ERR_UNSUPPORTED, "ViT width %d unsupported", cfg.pv.D);
  MD_HIP(hipSetDevice(dev->ordinal));
  md_model_s* m = new md_model_s();
  m->dev = dev;
  m->cfg = cfg;
  m->prec = cfg.precision;
  m->esz = cfg.precision == MD_PREC_F32 ? 4 : 2;
  m->ke = 128 / m->esz;
  m->xm = cfg.precision == MD_PREC_F16X2 ? 2 : 1;
  m->wterms = m->xm == 2 ? 3 : 1;
  m->ngroups = 2 + ((cfg.use_fov_head && cfg.has_fov_vit) ? 1 : 0);
  m->ln_fold_can = (m->prec == MD_PREC_BF16 || m->prec == MD_PREC_F16 || m->prec == MD_PREC_F16X2) && cfg.pv.D == 1024;  // four 256-column tiles per row: the consumer reads exactly four partials
  m->S = cfg.img_size();
  m->win = cfg.pv.img;
  m->g = cfg.pv.grid();
  m->P = m->g * m->g;
  m->NT = m->P + 1;
  m->SS = round_up(m->NT, 4);
  m->kpad = round_up(m->NT, 64);
  split_geometry(m->S, m->win, 0.25f, &m->stride0, &m->steps0);      // encoder.rs:329
  split_geometry(m->S / 2, m->win, 0.5f, &m->stride1, &m->steps1);   // encoder.rs:330
  m->pad_hi = feature_padding(m->win, m->stride0, m->g);
  m->pad_mid = feature_padding(m->win, m->stride1, m->g);
  m->mh_hi = merged_extent(m->g, m->steps0, m->pad_hi);
  m->mh_mid = merged_extent(m->g, m->steps1, m->pad_mid);
  auto fail = [&](int code) {
    model_destroy(m);
    return code;
  };
  if ((m->steps0 - 1) * m->stride0 + m->win != m->S || (m->steps1 - 1) * m->stride1 + m->win != m->S / 2 ||
      m->stride0 % 4 != 0 || m->stride1 % 4 != 0 || m->g - 2 * m->pad_hi <= 0 || m->g - 2 * m->pad_mid <= 0) {
    set_error("split geometry of window %d at image %d is not tile-exact", m->win, m->S);
    return fail(MD_ERR_UNSUPPORTED);
  }
  if (m->mh_mid * 2 != m->mh_hi || m->g * 4 != m->mh_hi) {
    // decoder levels must nest by factors of two (encoder.rs:416-434 / decoder.rs:119-134)
    set_error("merged feature maps %d / %d / %d do not nest by 2", m->mh_hi, m->mh_mid, m->g);
    return fail(MD_ERR_UNSUPPORTED);
  }

  // ---- parameters: fp32 master arena ----
  m->params = depth_pro_param_specs(cfg, MD_INIT_REFERENCE);
  size_t off = 0;
  std::vector<size_t> offs;
  for (size_t i = 0; i < m->params.size(); ++i) {
    m->pindex[m->params[i].name] = (int)i;
    offs.push_back(off);
    off += align_up(m->params[i].count() * 4, 256);
  }
  m->w32_bytes = off;
  if (hipMalloc((void**)&m->w32_base, m->w32_bytes) != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the fp32 weights failed", m->w32_bytes);
    return fail(MD_ERR_OOM);
  }
  (void)hipMemset(m->w32_base, 0, m->w32_bytes);
  for (size_t i = 0; i < m->params.size(); ++i) m->w32.push_back((float*)(m->w32_base + offs[i]));

  // ---- pack plan ----
  const int D = cfg.pv.D, F = cfg.F;
  const int* dims = cfg.pv.feat_dims;
  const char* vnames[3] = {"encoder.patch_encoder", "encoder.image_encoder", "fov.encoder"};
  for (int gi = 0; gi < m->ngroups; ++gi) {
    const std::string v = vnames[gi];
    add_pack(m, v + ".patch_embed.proj.weight", PACK_NK, D, 3 * cfg.pv.ps * cfg.pv.ps, 1);
    for (int i = 0; i < cfg.pv.depth; ++i) {
      const std::string b = v + ".blocks." + std::to_string(i);
      add_pack(m, b + ".attn.qkv.weight", PACK_NK, 3 * D, D, 1);
      add_pack(m, b + ".attn.proj.weight", PACK_NK, D, D, 1);
      add_pack(m, b + ".mlp.fc1.weight", PACK_NK, 4 * D, D, 1);
      add_pack(m, b + ".mlp.fc2.weight", PACK_NK, D, 4 * D, 1);
    }
  }
  auto pub = [&](const std::string& n, int din, int dout, int layers, int dint) {
    const int inter = dint > 0 ? dint : dout;
    add_pack(m, n + ".projection.weight", PACK_NK, inter, din, 1);
    for (int l = 0; l < layers; ++l)
      add_pack(m, n + ".upsample." + std::to_string(l) + ".weight", PACK_DECONV, l == 0 ? inter : dout, dout, 2);
  };
  pub("encoder.upsample_latent0", D, F, 3, dims[0]);
  pub("encoder.upsample_latent1", D, dims[0], 2, 0);
  // the last two k2s2 deconvolutions of each latent chain run as one k4s4 (their 2x intermediate is never written)
  add_pack_deconv_pair(m, "encoder.upsample_latent0.upsample.1x2", "encoder.upsample_latent0.upsample.1.weight",
                       "encoder.upsample_latent0.upsample.2.weight", F, F, F);
  add_pack_deconv_pair(m, "encoder.upsample_latent1.upsample.0x1", "encoder.upsample_latent1.upsample.0.weight",
                       "encoder.upsample_latent1.upsample.1.weight", dims[0], dims[0], dims[0]);
  pub("encoder.upsample0", D, dims[1], 1, 0);
  pub("encoder.upsample1", D, dims[2], 1, 0);
  pub("encoder.upsample2", D, dims[3], 1, 0);
  add_pack(m, "encoder.upsample_lowres.weight", PACK_DECONV, cfg.iv.D, dims[3], 2);
  add_pack(m, "encoder.fuse_lowres.weight", PACK_NK, dims[3], 2 * dims[3], 1);
  const int ddims[5] = {F, dims[0], dims[1], dims[2], dims[3]};
  for (int l = 1; l < 5; ++l) add_pack(m, "decoder.convs." + std::to_string(l) + ".conv.weight", PACK_CONV3, F, ddims[l], 3);
  for (int l = 0; l < 5; ++l) {
    const std::string f = "decoder.fusions." + std::to_string(l);
    for (const char* r : {"resnet1", "resnet2"}) {
      add_pack(m, f + "." + r + ".conv1.weight", PACK_CONV3, F, F, 3);
      add_pack(m, f + "." + r + ".conv2.weight", PACK_CONV3, F, F, 3);
    }
    if (l != 0)
      add_pack_composed(m, f + ".deconv_out_conv", f + ".deconv.weight", f + ".out_conv.weight", F, F);
    else
      add_pack(m, f + ".out_conv.weight", PACK_NK, F, F, 1);
  }
  add_pack(m, "head.conv0.weight", PACK_CONV3, F / 2, F, 3);
  add_pack_c1c3(m, "head.outconv_conv0", "decoder.fusions.0.out_conv", "head.conv0", F, F, F / 2);
  add_pack(m, "head.deconv.weight", PACK_DECONV, F / 2, F / 2, 2);
  add_pack_head_fused(m, "head.deconv_conv1", "head.deconv", "head.conv1", F / 2, F / 2, 32);
  add_pack(m, "head.conv1.weight", PACK_CONV3, 32, F / 2, 3);  // head_debug's un-fused conv1 (mod.rs:292); 74 KB
  if (cfg.use_fov_head) {
    if (cfg.has_fov_vit) {
      add_pack(m, "fov.encoder_proj.weight", PACK_NK, F / 2, cfg.fv.D, 1);
      add_pack(m, "fov.downsample_blocks.0.conv.weight", PACK_DIRECT, F / 2, F, 3, true);
      // the same weight as an implicit-GEMM operand: the stride-2 downsample of the lowres feature (fov.rs:79-87,185) is 2.7 GFLOP at B = 8 and
      // took 374 us as a direct convolution (one wave per output pixel) -- the MFMA family's stride-2 3x3 form does it in a tenth of that
      add_pack_as(m, "fov.downsample_blocks.0.conv.gemm", "fov.downsample_blocks.0.conv.weight", PACK_CONV3, F / 2, F, 3);
      add_pack(m, "fov.head_blocks.0.conv.weight", PACK_DIRECT, F / 4, F / 2, 3, true);
      add_pack(m, "fov.head_blocks.1.conv.weight", PACK_DIRECT, F / 8, F / 4, 3, true);
      add_pack(m, "fov.head_blocks.2.conv.weight", PACK_DIRECT, 1, F / 8, 6, true);
    } else {
      add_pack(m, "fov.head_blocks.0.conv.weight", PACK_DIRECT, F / 2, F, 3, true);
      add_pack(m, "fov.head_blocks.1.conv.weight", PACK_DIRECT, F / 4, F / 2, 3, true);
      add_pack(m, "fov.head_blocks.2.conv.weight", PACK_DIRECT, F / 8, F / 4, 3, true);
      add_pack(m, "fov.head_blocks.3.conv.weight", PACK_DIRECT, 1, F / 8, 6, true);
    }
  }
  size_t poff = 0;
  for (auto& e : m->packs) {
    e.dst = (void*)poff;  // offset for now
    poff += align_up(e.bytes + 256, 256);
  }
  m->wpk_bytes = poff;
  if (hipMalloc((void**)&m->wpk_base, m->wpk_bytes) != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the packed weights failed", m->wpk_bytes);
    return fail(MD_ERR_OOM);
  }
  (void)hipMemset(m->wpk_base, 0, m->wpk_bytes);
  for (auto& e : m->packs) e.dst = m->wpk_base + (size_t)e.dst;

  // ---- ViT weight tables ----
  for (int gi = 0; gi < m->ngroups; ++gi) {
    const std::string v = vnames[gi];
    VitW& w = m->vit[gi];
    w.pe_w = PK(m, v + ".patch_embed.proj.weight");
    w.pe_b = P32(m, v + ".patch_embed.proj.bias");
    w.cls = P32(m, v + ".cls_token");
    w.pos = P32(m, v + ".pos_embed");
    w.norm_g = P32(m, v + ".norm.gamma");
    w.norm_b = P32(m, v + ".norm.beta");
    for (int i = 0; i < cfg.pv.depth; ++i) {
      const std::string b = v + ".blocks." + std::to_string(i);
      VitBlockW k;
      k.n1g = P32(m, b + ".norm1.gamma"); k.n1b = P32(m, b + ".norm1.beta");
      k.n2g = P32(m, b + ".norm2.gamma"); k.n2b = P32(m, b + ".norm2.beta");
      k.qkv_w = PK(m, b + ".attn.qkv.weight"); k.qkv_b = P32(m, b + ".attn.qkv.bias");
      k.proj_w = PK(m, b + ".attn.proj.weight"); k.proj_b = P32(m, b + ".attn.proj.bias");
      k.ls1 = P32(m, b + ".ls1.gamma"); k.ls2 = P32(m, b + ".ls2.gamma");
      k.fc1_w = PK(m, b + ".mlp.fc1.weight"); k.fc1_b = P32(m, b + ".mlp.fc1.bias");
      k.fc2_w = PK(m, b + ".mlp.fc2.weight"); k.fc2_b = P32(m, b + ".mlp.fc2.bias");
      w.blk.push_back(k);
    }
  }
  if (m->ln_fold_can) {  // c / d of every block's two folded LayerNorms: [group][block][qkv_c 3D | qkv_d 3D | fc1_c 4D | fc1_d 4D]
    const size_t per_blk = (size_t)14 * D;
    if (hipMalloc((void**)&m->lnfold_base, ((size_t)m->ngroups * cfg.pv.depth * per_blk + 4 * D) * 4) != hipSuccess ||
        hipMemset(m->lnfold_base + (size_t)m->ngroups * cfg.pv.depth * per_blk, 0, (size_t)4 * D * 4) != hipSuccess) {  // (+ 4D zeros: the diagnostic form's c)
      set_error("hipMalloc of the LayerNorm-fold vectors failed");
      return fail(MD_ERR_OOM);
    }
    for (int gi = 0; gi < m->ngroups; ++gi)
      for (int i = 0; i < cfg.pv.depth; ++i) {
        float* q = m->lnfold_base + ((size_t)gi * cfg.pv.depth + i) * per_blk;
        VitBlockW& k = m->vit[gi].blk[i];
        k.qkv_c = q; k.qkv_d = q + 3 * D; k.fc1_c = q + 6 * D; k.fc1_d = q + 10 * D;
      }
  }

  // ---- workspace ----
  m->buf = new md_model_s::Buffers();
  size_t need = 0;
  plan_workspace(m, true, &need);
  if (hipMalloc((void**)&m->ws.base, need) != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the workspace failed (max_batch=%d)", need, cfg.max_batch);
    return fail(MD_ERR_OOM);
  }
  m->ws.cap = need;
  if (hipMemset(m->ws.base, 0, need) != hipSuccess) {  // padding rows/channels/keys must be finite zeros
    set_error("hipMemset of the workspace failed");
    return fail(MD_ERR_HIP);
  }
  int st = plan_workspace(m, false, nullptr);
  if (st != MD_OK) return fail(st);
  if (m->prec == MD_PREC_BF16 && (st = attention_asm_prepare()) != MD_OK) return fail(st);  // the code object loads here, never inside a capture
  if (hipMalloc(&m->zero_page, 4096) != hipSuccess) return fail(MD_ERR_OOM);
  (void)hipMemset(m->zero_page, 0, 4096);
  (void)hipDeviceSynchronize();
  *out = m;
  return MD_OK;
}

int model_destroy(md_model_t m) {
  if (!m) return MD_OK;
  if (m->forks.load() > 0) MD_FAIL(MD_ERR_INVALID_ARG, "model has %d live fork(s) sharing its weights: destroy them first", m->forks.load());
  if (m->dev) (void)hipSetDevice(m->dev->ordinal);
  (void)hipDeviceSynchronize();
  if (m->parent) {  // a fork owns no weights
    m->parent->forks.fetch_sub(1);
    m->w32_base = nullptr;
    m->wpk_base = nullptr;
  }
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  if (m->w32_base) (void)hipFree(m->w32_base);
  if (m->wpk_base) (void)hipFree(m->wpk_base);
  if (m->lnfold_base && !m->parent) (void)hipFree(m->lnfold_base);
  if (m->ws.base) (void)hipFree(m->ws.base);
  if (m->zero_page) (void)hipFree(m->zero_page);
  for (auto& kv : m->index_tables) (void)hipFree(kv.second);
  for (auto& kv : m->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  for (auto& kv : m->taps) (void)hipFree(kv.second.dev);
  for (auto& t : m->timing) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  if (m->da3) da3_destroy_state(m);
  if (m->buf) {
    if (m->buf->xraw) (void)hipFree(m->buf->xraw);
    if (m->buf->rgb) (void)hipFree(m->buf->rgb);
    if (m->buf->depth_stage) (void)hipFree(m->buf->depth_stage);
    if (m->buf->pin_in) (void)hipHostFree(m->buf->pin_in);
    if (m->buf->pin_out) (void)hipHostFree(m->buf->pin_out);
    delete m->buf;
  }
  delete m;
  return MD_OK;
}

// `DepthPro` is `Module + Clone` and `infer(&self)` takes a shared reference (depth_pro/mod.rs:119-126,312;
// crates/bevy_burn_depth/src/lib.rs:18,29): a second in-flight inference needs a second workspace, not a second copy
// of the 5.6 GB of weights. The fork aliases the root's fp32 parameter arena and packed MFMA operand arena.
int model_fork(md_model_t src, md_model_t* out) {
  if (!src || !out) MD_FAIL(MD_ERR_INVALID_ARG, "model/out is null");
  if (src->kind != 0)
    MD_FAIL(MD_ERR_UNSUPPORTED, "md_model_fork: Depth-Anything-v3 models keep per-shape tables beside their workspace "
                                "(the reference's CachedDepthAnything3 is not Sync either, depth_anything3/mod.rs:44,67-70)");
  md_model_s* root = model_root(src);
  if (!root->committed) MD_FAIL(MD_ERR_INVALID_ARG, "weights were modified; call md_model_commit_weights before forking");
  MD_HIP(hipSetDevice(root->dev->ordinal));
  md_model_s* m = new md_model_s();
  m->dev = root->dev;
  m->cfg = root->cfg;
  m->prec = root->prec; m->esz = root->esz; m->ke = root->ke; m->xm = root->xm; m->wterms = root->wterms;
  m->params = root->params; m->pindex = root->pindex; m->w32 = root->w32;
  m->w32_base = root->w32_base; m->w32_bytes = root->w32_bytes;
  m->packs = root->packs; m->pack_index = root->pack_index;
  m->wpk_base = root->wpk_base; m->wpk_bytes = root->wpk_bytes;
  m->committed = true;
  m->ngroups = root->ngroups;
  m->ln_fold_can = root->ln_fold_can; m->ln_fold_opt = root->ln_fold_opt; m->lnfold_base = root->lnfold_base;
  for (int g = 0; g < 3; ++g) m->vit[g] = root->vit[g];
  m->head_b_host = root->head_b_host;
  m->S = root->S; m->win = root->win; m->g = root->g; m->P = root->P; m->NT = root->NT; m->SS = root->SS; m->kpad = root->kpad;
  m->steps0 = root->steps0; m->stride0 = root->stride0; m->steps1 = root->steps1; m->stride1 = root->stride1;
  m->pad_hi = root->pad_hi; m->pad_mid = root->pad_mid; m->mh_hi = root->mh_hi; m->mh_mid = root->mh_mid;
  m->parent = root;
  root->forks.fetch_add(1);
  auto fail = [&](int code) {
    model_destroy(m);
    return code;
  };
  // same flags as the root's default stream (md_device_open): with stream == NULL either context is ordered against work
  // the caller left on the legacy null stream; two contexts still run concurrently with each other
  if (hipStreamCreateWithFlags(&m->own_stream, hipStreamDefault) != hipSuccess) {
    set_error("hipStreamCreate failed");
    return fail(MD_ERR_HIP);
  }
  m->buf = new md_model_s::Buffers();
  size_t need = 0;
  plan_workspace(m, true, &need);
  if (hipMalloc((void**)&m->ws.base, need) != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the fork's workspace failed (max_batch=%d)", need, m->cfg.max_batch);
    return fail(MD_ERR_OOM);
  }
  m->ws.cap = need;
  if (hipMemset(m->ws.base, 0, need) != hipSuccess) return fail(MD_ERR_HIP);  // padding rows / channels / keys: finite zeros
  int st = plan_workspace(m, false, nullptr);
  if (st != MD_OK) return fail(st);
  if (m->prec == MD_PREC_BF16 && (st = attention_asm_prepare()) != MD_OK) return fail(st);  // the code object loads here, never inside a capture
  if (hipMalloc(&m->zero_page, 4096) != hipSuccess) return fail(MD_ERR_OOM);
  (void)hipMemset(m->zero_page, 0, 4096);
  (void)hipDeviceSynchronize();
  *out = m;
  return MD_OK;
}

int model_init_seeded(md_model_t m, uint64_t seed, int scheme) {
  if (scheme != MD_INIT_REFERENCE && scheme != MD_INIT_PARITY) MD_FAIL(MD_ERR_INVALID_ARG, "unknown init scheme %d", scheme);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  std::vector<ParamSpec> specs = depth_pro_param_specs(m->cfg, scheme);
  if (specs.size() != m->params.size()) MD_FAIL(MD_ERR_FORMAT, "internal: inventory mismatch");
  std::vector<float> tmp;
  for (size_t i = 0; i < specs.size(); ++i) {
    const size_t n = specs[i].count();
    tmp.resize(n);
    uniform_stream(specs[i].name, seed, n, specs[i].lo, specs[i].hi, tmp.data());
    MD_HIP(hipMemcpy(m->w32[i], tmp.data(), n * 4, hipMemcpyHostToDevice));
  }
  return model_commit(m);
}

int model_load_params_from_container(md_model_t m, const char* path) {
  Container c;
  MD_TRY(read_container(path, &c));
  MD_HIP(hipSetDevice(m->dev->ordinal));
  std::vector<float> tmp;
  for (size_t i = 0; i < m->params.size(); ++i) {
    const ParamSpec& s = m->params[i];
    auto it = c.tensors.find(s.name);
    if (it == c.tensors.end()) MD_FAIL(MD_ERR_FORMAT, "checkpoint `%s` has no tensor `%s`", path, s.name.c_str());
    size_t n = 1;
    for (auto d : it->second.shape) n *= (size_t)d;
    if (n != s.count()) {
      MD_FAIL(MD_ERR_FORMAT, "tensor `%s` has %zu elements, the model expects %zu", s.name.c_str(), n, s.count());
    }
    // A Burn record stores `nn::Linear` weights [d_input, d_output] (Container::burn_record); the inventory -- and the
    // MFMA pack -- hold them [out, in] like PyTorch. Every rank-2 parameter of the inventory is such a weight.
    bool linear_t = false;
    if (c.burn_record && s.shape.size() == 2 && it->second.shape.size() == 2) {
      if (it->second.shape[0] != s.shape[1] || it->second.shape[1] != s.shape[0])
        MD_FAIL(MD_ERR_FORMAT, "Burn record: linear weight `%s` has shape [%lld, %lld], expected [d_input = %lld, d_output = %lld]", s.name.c_str(),
                (long long)it->second.shape[0], (long long)it->second.shape[1], (long long)s.shape[1], (long long)s.shape[0]);
      linear_t = true;
    }
    // ConvTranspose weights may arrive [out,in,..] (maybe_fix_conv_transpose2d, mod.rs:416-431)
    bool swap = false;
    if (s.shape.size() == 4 && it->second.shape.size() == 4 && s.shape[2] == 2 && s.shape[3] == 2 &&
        s.shape[0] != s.shape[1] && it->second.shape[0] == s.shape[1] && it->second.shape[1] == s.shape[0])
      swap = true;
    tmp.resize(n);
    MD_TRY(container_tensor_to_f32(c, it->second, tmp.data(), n));
    if (linear_t) {  // [in, out] -> [out, in]
      std::vector<float> t2(n);
      const size_t no = (size_t)s.shape[0], ni = (size_t)s.shape[1];
      for (size_t ii = 0; ii < ni; ++ii)
        for (size_t o = 0; o < no; ++o) t2[o * ni + ii] = tmp[ii * no + o];
      tmp.swap(t2);
    }
    if (swap) {
      std::vector<float> t2(n);
      const size_t a = (size_t)s.shape[0], b2 = (size_t)s.shape[1];
      for (size_t o = 0; o < b2; ++o)
        for (size_t ii = 0; ii < a; ++ii)
          for (int q = 0; q < 4; ++q) t2[(ii * b2 + o) * 4 + q] = tmp[(o * a + ii) * 4 + q];
      tmp.swap(t2);
    }
    MD_HIP(hipMemcpy(m->w32[i], tmp.data(), n * 4, hipMemcpyHostToDevice));
  }
  return MD_OK;
}

int model_load_container(md_model_t m, const char* path) {
  MD_TRY(model_load_params_from_container(m, path));
  return model_commit(m);
}

int model_commit(md_model_t m) {
  if (m->parent) MD_FAIL(MD_ERR_INVALID_ARG, "a fork shares its root's weights: commit on the root model");
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t s = m->dev->stream;
  float* composed = nullptr;  // fp32 staging of a composed weight
  size_t composed_elems = 0;
  for (auto& e : m->packs) {
    if (e.kind == PACK_HEAD_W) composed_elems = std::max(composed_elems, (size_t)4 * e.d0 * e.d1 * 9);
    else if (e.kind == PACK_C1C3_W) composed_elems = std::max(composed_elems, (size_t)e.d0 * e.d1 * 9);
    else if (e.kind == PACK_HEAD_B || e.kind == PACK_C1C3_B) continue;
    else if (e.param2 >= 0) composed_elems = std::max(composed_elems, (size_t)e.d0 * e.d1 * e.k * e.k);
  }
  if (composed_elems) MD_HIP(hipMalloc((void**)&composed, composed_elems * sizeof(float)));
  if (m->xm == 2) {
    // split-half operands: a product with an f16-exact weight is two MFMA terms, else three. One form per model (the three
    // ViTs share launches): two when EVERY plain weight is exact -- an f16 checkpoint -- else three. The layer products
    // composed at commit (fp32 sums of products) are never exact: always three.
    unsigned* d_cnt = nullptr;
    unsigned h_cnt = 0;
    MD_HIP(hipMalloc((void**)&d_cnt, 4));
    MD_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
    for (auto& e : m->packs) {
      if (e.f32 || e.param2 >= 0 || (e.kind != PACK_NK && e.kind != PACK_CONV3 && e.kind != PACK_DECONV)) continue;
      const long n = (long)m->params[e.param].count();
      hipLaunchKernelGGL(count_inexact_f16_kernel, dim3((int)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, s, m->w32[e.param], n, d_cnt);
    }
    MD_HIP(hipMemcpyAsync(&h_cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
    MD_HIP(hipStreamSynchronize(s));
    MD_HIP(hipFree(d_cnt));
    m->wterms = h_cnt == 0 ? 2 : 3;
    for (auto& e : m->packs) e.terms = e.f32 ? 1 : ((e.param2 >= 0 || e.kind == PACK_HEAD_W || e.kind == PACK_C1C3_W) ? 3 : m->wterms);
  }
  for (auto& e : m->packs) {
    if (e.kind == PACK_HEAD_W) {  // deconv -> conv3x3 of the depth head as one conv (d0 = Cout, d1 = Cin, k = Cmid)
      MD_TRY(compose_head(m->w32[e.param], m->w32[e.param2], e.d1, e.k, e.d0, composed, s));
      PackEntry c3 = e;
      c3.kind = PACK_CONV3; c3.d0 = 4 * e.d0; c3.k = 3;
      MD_TRY(pack_weight(composed, c3, m->prec, s));
    } else if (e.kind == PACK_HEAD_B || e.kind == PACK_C1C3_B) {
      MD_TRY(compose_head_bias(m->w32[e.param2], m->w32[e.param3], m->w32[e.param4], e.k, e.d0, (float*)e.dst, s));
    } else if (e.kind == PACK_C1C3_W) {  // d0 = Cout, d1 = Cin, k = Cmid
      MD_TRY(compose_c1c3(m->w32[e.param], m->w32[e.param2], e.d1, e.k, e.d0, composed, s));
      PackEntry c3 = e;
      c3.kind = PACK_CONV3; c3.k = 3;
      MD_TRY(pack_weight(composed, c3, m->prec, s));
    } else if (e.param2 < 0) {
      MD_TRY(pack_weight(m->w32[e.param], e, m->prec, s));
    } else if (e.k == 4) {  // deconv k2s2 -> deconv k2s2 as one k4s4 (d0 = Cin, d1 = Cout, param3 = Cmid)
      MD_TRY(compose_deconv_pair(m->w32[e.param], m->w32[e.param2], e.d0, e.param3, e.d1, composed, s));
      MD_TRY(pack_weight(composed, e, m->prec, s));
    } else {
      MD_TRY(compose_deconv_conv(m->w32[e.param], m->w32[e.param2], e.d0, e.d1, e.d1, composed, s));
      MD_TRY(pack_weight(composed, e, m->prec, s));
    }
  }
  if (m->kind == 0 && m->ln_fold_can && m->lnfold_base) {
    const int D = m->cfg.pv.D;
    const char* vnames[3] = {"encoder.patch_encoder", "encoder.image_encoder", "fov.encoder"};
    for (int gi = 0; gi < m->ngroups; ++gi)
      for (int i = 0; i < m->cfg.pv.depth; ++i) {
        const std::string b = std::string(vnames[gi]) + ".blocks." + std::to_string(i);
        const VitBlockW& k = m->vit[gi].blk[i];
        MD_TRY(launch_ln_fold_vectors(P32(m, b + ".attn.qkv.weight"), k.n1g, k.n1b, k.qkv_b, 3 * D, D, m->prec, (float*)k.qkv_c, (float*)k.qkv_d, s));
        MD_TRY(launch_ln_fold_vectors(P32(m, b + ".mlp.fc1.weight"), k.n2g, k.n2b, k.fc1_b, 4 * D, D, m->prec, (float*)k.fc1_c, (float*)k.fc1_d, s));
      }
  }
  auto it = m->pindex.find(m->kind == 1 ? "head_mono.scratch.output_conv2.conv2.bias" : "head.conv_out.bias");
  if (it != m->pindex.end()) MD_HIP(hipMemcpyAsync(&m->head_b_host, m->w32[it->second], 4, hipMemcpyDeviceToHost, s));
  MD_HIP(hipStreamSynchronize(s));
  if (composed) MD_HIP(hipFree(composed));
  if (m->kind == 1) MD_TRY(da3_on_commit(m));
  // captured graphs bake by-value launch parameters (the head's output bias, the split-half term count): none survives a commit
  for (auto& kv : m->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  m->graphs.clear();
  m->commit_gen += 1;
  m->committed = true;
  return MD_OK;
}

// `DepthPro::load` reads an f16 record (`HalfPrecisionSettings`, depth_pro/mod.rs:206): every parameter of a loaded reference
// model is an IEEE half widened to f32. This rounds the fp32 master copy of every parameter the same way, in place.
int model_round_weights_f16(md_model_t m) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  if (m->parent) MD_FAIL(MD_ERR_INVALID_ARG, "a fork shares its root's weights: round them on the root model");
  if (m->forks.load() > 0) MD_FAIL(MD_ERR_INVALID_ARG, "model has %d live fork(s) sharing its weights", m->forks.load());
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t s = m->dev->stream;
  for (size_t i = 0; i < m->params.size(); ++i) {
    const long n = (long)m->params[i].count();
    hipLaunchKernelGGL(round_f16_kernel, dim3((int)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, s, m->w32[i], n);
  }
  MD_HIP(hipGetLastError());
  MD_HIP(hipStreamSynchronize(s));
  m->committed = false;
  return model_commit(m);
}

// ------------------------------------------------------------------------------------------------
// index tables (token -> merged map gather, encoder.rs:234-319)
// ------------------------------------------------------------------------------------------------
static int get_index_set(md_model_s* m, int B, md_model_s::IndexSet* out) {
  auto it = m->index_sets.find(B);
  if (it != m->index_sets.end()) {
    *out = it->second;
    return MD_OK;
  }
  const int g = m->g, SS = m->SS, P = m->P;
  const int n0 = m->steps0 * m->steps0 * B, n1 = m->steps1 * m->steps1 * B;
  const int nseq_p = n0 + n1 + B;
  const size_t nhi = (size_t)B * m->mh_hi * m->mh_hi, nmid = (size_t)B * m->mh_mid * m->mh_mid, nlo = (size_t)B * P;
  std::vector<int> h(nhi + nmid + 3 * nlo);
  size_t o = 0;
  for (int b = 0; b < B; ++b)
    for (int Y = 0; Y < m->mh_hi; ++Y)
      for (int X = 0; X < m->mh_hi; ++X) {
        int j, i, ty, tx;
        merge_source(Y, X, g, g, m->steps0, m->pad_hi, &j, &i, &ty, &tx);
        h[o++] = ((j * m->steps0 + i) * B + b) * SS + 1 + ty * g + tx;
      }
  for (int b = 0; b < B; ++b)
    for (int Y = 0; Y < m->mh_mid; ++Y)
      for (int X = 0; X < m->mh_mid; ++X) {
        int j, i, ty, tx;
        merge_source(Y, X, g, g, m->steps1, m->pad_mid, &j, &i, &ty, &tx);
        h[o++] = (n0 + (j * m->steps1 + i) * B + b) * SS + 1 + ty * g + tx;
      }
  for (int which = 0; which < 3; ++which)
    for (int b = 0; b < B; ++b)
      for (int p = 0; p < P; ++p) {
        const int seq = which == 0 ? n0 + n1 + b : which == 1 ? nseq_p + b : nseq_p + B + b;
        h[o++] = seq * SS + 1 + p;
      }
  int* d = nullptr;
  MD_HIP(hipMalloc((void**)&d, h.size() * 4));
  m->alloc_count += 1;
  MD_HIP(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  m->index_tables[B] = d;
  md_model_s::IndexSet s;
  s.hi = d;
  s.mid = d + nhi;
  s.x2 = s.mid + nmid;
  s.img = s.x2 + nlo;
  s.fov = s.img + nlo;
  m->index_sets[B] = s;
  *out = s;
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// forward schedule
// ------------------------------------------------------------------------------------------------
// The ViT stage over the sequences [s_lo, s_hi) of the 37 B (patch tiles, image, fov): the whole range in the ordinary
// call; one rank's share in the tile-parallel mode (SURVEY 8(e) second mode -- the tiles of encoder.rs:329-348 never
// interact before `merge`, so any split of the sequence range is exact). Every launch addresses absolute rows of the
// workspace, so the windows of different ranks (or of successive calls) fill disjoint row ranges of the same buffers.
static int run_vit(Run& r, int nseq_p, int nseq, int s_lo, int s_hi) {
  md_model_s* m = r.m;
  md_model_s::Buffers* b = m->buf;
  const ModelCfg& c = m->cfg;
  const int B = r.B, D = c.pv.D, P = m->P, SS = m->SS, NT = m->NT, heads = c.pv.heads;
  const int n0 = m->steps0 * m->steps0 * B, n1 = m->steps1 * m->steps1 * B;
  const int Kpe = 3 * c.pv.ps * c.pv.ps;
  const int gseq0[3] = {0, nseq_p, nseq_p + B};
  const int gnseq[3] = {nseq_p, B, B};
  if (s_lo < 0 || s_hi > nseq || s_lo > s_hi) MD_FAIL(MD_ERR_INVALID_ARG, "sequence window [%d, %d) of %d", s_lo, s_hi, nseq);
  const int WS = s_hi - s_lo;  // sequences of this window
  if (WS == 0) return MD_OK;
  // encoder groups clipped to the window: slot g of the launches is encoder gi[g] on sequences [glo[g], glo[g] + gcnt[g])
  int G = 0, gi[3], glo[3], gcnt[3];
  for (int g = 0; g < m->ngroups; ++g) {
    const int lo = std::max(gseq0[g], s_lo), hi = std::min(gseq0[g] + gnseq[g], s_hi);
    if (hi > lo) { gi[G] = g; glo[G] = lo; gcnt[G] = hi - lo; ++G; }
  }
  const size_t esz = (size_t)m->esz * m->xm;  // bytes per logical element of a T tensor
  auto trow = [&](void* base, int width) { return (void*)((char*)base + (size_t)s_lo * SS * width * esz); };
  float* xres_w = b->xres + (size_t)s_lo * SS * D;

  SeqGroups sg;  // sequence numbers relative to the window
  sg.ngroups = G;
  for (int g = 0; g < 4; ++g) { sg.seq0[g] = 0; sg.nseq[g] = 0; sg.a[g] = nullptr; sg.b[g] = nullptr; }
  for (int g = 0; g < G; ++g) { sg.seq0[g] = glo[g] - s_lo; sg.nseq[g] = gcnt[g]; }

  // cls + pos[0], padding rows
  for (int g = 0; g < G; ++g) { sg.a[g] = m->vit[gi[g]].cls; sg.b[g] = m->vit[gi[g]].pos; }
  r.begin("cls_init");
  MD_TRY(launch_cls_init(xres_w, WS, SS, NT, D, sg, r.st));
  r.end();

  // patch embed (conv 16x16 s16 as GEMM, + bias + pos_embed)
  {
    GemmParams p;
    p.N = D; p.K = Kpe; p.ngroups = G;
    for (int g = 0; g < G; ++g) {
      p.g_row0[g] = glo[g] * P;
      p.g_rows[g] = gcnt[g] * P;
      // image/fov encoders read the x2 tiles (encoder.rs:409, fov.rs:202)
      p.g_arow0[g] = gi[g] == 0 ? glo[g] * P : (n0 + n1 + (glo[g] - gseq0[gi[g]])) * P;
      p.W[g] = m->vit[gi[g]].pe_w;
      p.bias[g] = m->vit[gi[g]].pe_b;
      p.pos[g] = m->vit[gi[g]].pos;
    }
    p.A = b->patches;
    split_dense_a(m, p, Kpe, Kpe, 0);
    p.epi = EPI_PATCH_EMBED; p.out = b->xres; p.ldo = D; p.seq_stride = SS; p.seq_patches = P; p.embed = D;
    r.begin("patch_embed");
    MD_TRY(launch_gemm(p, A_DENSE, m->prec, TILE_AUTO, r.st));
    r.end();
  }
  const long rows = (long)WS * SS;
  auto group_rows = [&](GemmParams& p) {
    p.ngroups = G;
    for (int g = 0; g < G; ++g) {
      p.g_row0[g] = glo[g] * SS;
      p.g_arow0[g] = glo[g] * SS;
      p.g_rows[g] = gcnt[g] * SS;
    }
  };
  const size_t vt_seq = (size_t)heads * 64 * m->kpad * m->esz;  // bytes of one sequence in a V^T plane
  // LayerNorm fold (gemm.h GemmParams::ln_*): norm2 of every block and norm1 of blocks 1.. never run as launches -- the GEMM that
  // produces the residual stream (proj / fc2) also writes round_T(gamma . x) and the rows' statistics, the GEMM behind the norm
  // (fc1 / qkv) finishes its accumulators with them. Block 0's norm1 (behind the patch embedding) and the final norm stay launches.
  // The four GEMMs of a block then always run the 256 x 256 kernel: the fold is a model-level choice, so that any window of a call
  // (tile-parallel mode) computes the same bits as the whole call.
  const bool fold = m->ln_fold_on();
  const int fold_tile = fold ? TILE_256x256 : TILE_AUTO;
  // ln_fold = 3 (diagnostic, benches only): the UNFOLDED schedule through the fold-form consumer kernels (EK 6 / 7) on neutral statistics
  // (rstd = 1, mu = 0, c = 0, d = bias): isolates what those epilogues cost from what the colder A operand of the folded schedule costs
  const bool neutral = !fold && m->ln_fold_can && m->ln_fold_opt == 3;
  const float* zero_c = m->lnfold_base ? m->lnfold_base + (size_t)m->ngroups * c.pv.depth * 14 * D : nullptr;
  if (neutral) MD_TRY(launch_fill_pairs(b->ln_ab + (size_t)s_lo * SS * 2, rows, 1.0f, 0.0f, r.st));
  auto fold_producer = [&](GemmParams& p, int blk, bool norm2) {
    p.ln_out = b->xn; p.ln_ldo = (long)D * m->xm; p.ln_plane = m->xm == 2 ? D : 0;
    p.ln_stats_out = b->ln_stats; p.ln_parts = D / 256;
    for (int g = 0; g < G; ++g) p.ln_gamma[g] = norm2 ? m->vit[gi[g]].blk[blk].n2g : m->vit[gi[g]].blk[blk].n1g;
  };
  // (ln_fold = 4: the pairs come from the ln_finish launch -- the A/B form; default: the consumer combines the partials itself)
  const bool finish_launch = neutral || m->ln_fold_opt == 4;
  auto fold_consumer = [&](GemmParams& p) {
    p.ln_stats = finish_launch ? b->ln_ab : b->ln_stats; p.ln_raw = finish_launch ? 0 : 1;
    p.ln_parts = D / 256; p.ln_inv_n = 1.0f / (float)D; p.ln_eps = c.ln_eps;
  };
  auto fold_finish = [&]() -> int {  // the window's rows: (mean, M2) x 4 -> (rstd, -mu rstd)
    if (!finish_launch) return MD_OK;
    r.begin("ln_finish");
    const int st_ = launch_ln_finish(b->ln_stats + (size_t)s_lo * SS * (D / 256) * 2, b->ln_ab + (size_t)s_lo * SS * 2, rows, 1.0f / (float)D, c.ln_eps, r.st);
    r.end();
    return st_;
  };
  for (int i = 0; i < c.pv.depth; ++i) {
    const bool fold1 = fold && i > 0;  // this block's norm1 was folded by the previous block's fc2
    if (!fold1) {
      for (int g = 0; g < G; ++g) { sg.a[g] = m->vit[gi[g]].blk[i].n1g; sg.b[g] = m->vit[gi[g]].blk[i].n1b; }
      r.begin("layernorm");
      MD_TRY(launch_layernorm(xres_w, trow(b->xn, D), rows, D, c.ln_eps, SS, sg, m->prec, 0, r.st));
      r.end();
    }
    {
      GemmParams p;
      p.N = 3 * D; group_rows(p);
      for (int g = 0; g < G; ++g) {
        p.W[g] = m->vit[gi[g]].blk[i].qkv_w;
        p.bias[g] = fold1 ? m->vit[gi[g]].blk[i].qkv_d : m->vit[gi[g]].blk[i].qkv_b;
        if (fold1) p.ln_c[g] = m->vit[gi[g]].blk[i].qkv_c;
      }
      if (fold1) fold_consumer(p);
      if (neutral) { fold_consumer(p); for (int g = 0; g < G; ++g) p.ln_c[g] = zero_c; }
      p.A = b->xn;
      split_dense_a(m, p, D, D, 0);
      p.v_plane = (long)m->vt_plane;
      p.epi = EPI_QKV; p.out = b->qk; p.vT = b->vT; p.seq_stride = SS; p.embed = D; p.heads = heads; p.kpad = m->kpad; p.qscale = attn_qscale(m->prec);
      r.begin("qkv_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, (fold1 || neutral) ? TILE_256x256 : TILE_AUTO, r.st));
      r.end();
    }
    void* vT_w = (char*)b->vT + (size_t)s_lo * vt_seq;
    if (m->prec != MD_PREC_F32) {
      r.begin("attention");
      MD_TRY(launch_attention(trow(b->qk, 2 * D), vT_w, trow(b->ao, D), WS, SS, NT, heads, D, m->kpad, m->prec, r.st, 0.f, (long)m->vt_plane,
                              b->attn_redo));
      r.end();
    } else {
      // fp32: scores = q k^T (batched GEMM) -> row softmax -> P V^T^T (batched GEMM)
      const float* qk_w = (const float*)trow(b->qk, 2 * D);
      GemmParams p;
      p.N = SS; p.K = 64; p.ngroups = 1; p.g_rows[0] = NT;
      p.batch = WS * heads; p.batch_inner = heads;
      p.A = qk_w; p.lda = 2 * D; p.a_bs[0] = (long)SS * 2 * D; p.a_bs[1] = 64;
      p.W[0] = qk_w + D; p.ldw = 2 * D; p.w_bs[0] = (long)SS * 2 * D; p.w_bs[1] = 64;
      p.epi = EPI_STORE; p.out_f32 = 1; p.out = b->scores; p.ldo = m->kpad;
      p.o_bs[0] = (long)heads * SS * m->kpad; p.o_bs[1] = (long)SS * m->kpad;
      r.begin("attn_scores_f32");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, TILE_128x128, r.st));
      r.end();
      r.begin("attn_softmax_f32");
      MD_TRY(launch_softmax_rows(b->scores, (long)WS * heads * SS, NT, m->kpad, 0.125f, r.st));
      r.end();
      GemmParams q;
      q.N = 64; q.K = m->kpad; q.ngroups = 1; q.g_rows[0] = NT;
      q.batch = WS * heads; q.batch_inner = heads;
      q.A = b->scores; q.lda = m->kpad; q.a_bs[0] = (long)heads * SS * m->kpad; q.a_bs[1] = (long)SS * m->kpad;
      q.W[0] = vT_w; q.ldw = m->kpad; q.w_bs[0] = (long)heads * 64 * m->kpad; q.w_bs[1] = 64L * m->kpad;
      q.epi = EPI_STORE; q.out = trow(b->ao, D); q.ldo = D; q.o_bs[0] = (long)SS * D; q.o_bs[1] = 64;
      r.begin("attn_pv_f32");
      MD_TRY(launch_gemm(q, A_DENSE, m->prec, TILE_128x128, r.st));
      r.end();
    }
    {
      GemmParams p;
      p.N = D; group_rows(p);
      for (int g = 0; g < G; ++g) {
        p.W[g] = m->vit[gi[g]].blk[i].proj_w; p.bias[g] = m->vit[gi[g]].blk[i].proj_b; p.scale[g] = m->vit[gi[g]].blk[i].ls1;
      }
      p.A = b->ao;
      split_dense_a(m, p, D, D, 0);
      p.epi = EPI_RESID_LS; p.out = b->xres; p.ldo = D;
      if (fold) fold_producer(p, i, true);
      r.begin("proj_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, fold_tile, r.st));
      r.end();
      if (fold) MD_TRY(fold_finish());
    }
    if (!fold) {
      for (int g = 0; g < G; ++g) { sg.a[g] = m->vit[gi[g]].blk[i].n2g; sg.b[g] = m->vit[gi[g]].blk[i].n2b; }
      r.begin("layernorm");
      MD_TRY(launch_layernorm(xres_w, trow(b->xn, D), rows, D, c.ln_eps, SS, sg, m->prec, 0, r.st));
      r.end();
    }
    {
      GemmParams p;
      p.N = 4 * D; group_rows(p);
      for (int g = 0; g < G; ++g) {
        p.W[g] = m->vit[gi[g]].blk[i].fc1_w;
        p.bias[g] = fold ? m->vit[gi[g]].blk[i].fc1_d : m->vit[gi[g]].blk[i].fc1_b;
        if (fold) p.ln_c[g] = m->vit[gi[g]].blk[i].fc1_c;
      }
      if (fold) fold_consumer(p);
      if (neutral) { fold_consumer(p); for (int g = 0; g < G; ++g) p.ln_c[g] = zero_c; }
      p.A = b->xn;
      split_dense_a(m, p, D, D, 0);
      p.epi = EPI_STORE; p.act = ACT_GELU; p.out = b->hbuf;
      split_out(m, p, 4 * D, true);
      r.begin("fc1_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, (fold || neutral) ? TILE_256x256 : TILE_AUTO, r.st));
      r.end();
    }
    {
      GemmParams p;
      p.N = D; group_rows(p);
      for (int g = 0; g < G; ++g) {
        p.W[g] = m->vit[gi[g]].blk[i].fc2_w; p.bias[g] = m->vit[gi[g]].blk[i].fc2_b; p.scale[g] = m->vit[gi[g]].blk[i].ls2;
      }
      p.A = b->hbuf;
      split_dense_a(m, p, 4 * D, 4 * D, 0);
      p.epi = EPI_RESID_LS; p.out = b->xres; p.ldo = D;
      if (fold && i + 1 < c.pv.depth) fold_producer(p, i + 1, false);
      r.begin("fc2_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, fold_tile, r.st));
      r.end();
      if (fold && i + 1 < c.pv.depth) MD_TRY(fold_finish());
    }
    // hooks: un-normalised tokens incl. cls after blocks hook_ids[0], hook_ids[1] (vit.rs:30,63): the first n0 sequences
    // (the 5 x 5 high-resolution tiles, encoder.rs:379-390)
    for (int hk = 0; hk < 2; ++hk)
      if (c.pv.hook_ids[hk] == i) {
        const int h_lo = std::min(s_lo, n0), h_hi = std::min(s_hi, n0);
        if (h_hi > h_lo) {
          r.begin("hook_copy");
          MD_TRY(launch_convert_rows(b->xres + (size_t)h_lo * SS * D, (char*)b->hook[hk] + (size_t)h_lo * SS * D * esz,
                                     (long)(h_hi - h_lo) * SS * D, m->prec, r.st, D));
          r.end();
        }
      }
  }
  for (int g = 0; g < G; ++g) { sg.a[g] = m->vit[gi[g]].norm_g; sg.b[g] = m->vit[gi[g]].norm_b; }
  r.begin("layernorm");
  MD_TRY(launch_layernorm(xres_w, trow(b->tok, D), rows, D, c.ln_eps, SS, sg, m->prec, 0, r.st));
  r.end();
  return MD_OK;
}

static int run_encoder_tail(Run& r, const md_model_s::IndexSet& ix) {
  md_model_s* m = r.m;
  md_model_s::Buffers* b = m->buf;
  const ModelCfg& c = m->cfg;
  const int B = r.B, D = c.pv.D, F = c.F, g = m->g;
  const int* dims = c.pv.feat_dims;
  const int hi = m->mh_hi, mid = m->mh_mid;
  const long Mhi = (long)B * hi * hi, Mmid = (long)B * mid * mid, Mlo = (long)B * g * g;
  auto W = [&](const char* n) { return PK(m, n); };
  // latent0: 1x1 (D -> dims0) then 3 deconvs -> F @ 8x (encoder.rs:146-151,423)
  MD_TRY(gemm_rows(r, "enc_proj", b->hook[0], D, ix.hi, Mhi, W("encoder.upsample_latent0.projection.weight"), dims[0], D,
                   nullptr, b->l0p, cpad(m, dims[0])));
  MD_TRY(deconv2(r, "enc_deconv", b->l0p, cpad(m, dims[0]), nullptr, hi, hi, W("encoder.upsample_latent0.upsample.0.weight"),
                 cpad(m, dims[0]), F, nullptr, b->l0a, cpad(m, F), 0));
  // upsample.1 and upsample.2 (k2s2, no bias, nothing between them) as ONE k4s4 deconvolution on their weight product
  MD_TRY(deconv2(r, "enc_deconv", b->l0a, cpad(m, F), nullptr, 2 * hi, 2 * hi, W("encoder.upsample_latent0.upsample.1x2"),
                 cpad(m, F), F, nullptr, b->enc0, cpad(m, F), 0, b->enc0r, 4, 3));
  // latent1: 1x1 (D -> dims0), 2 deconvs @ 4x (encoder.rs:152,424): the two deconvolutions as one k4s4
  MD_TRY(gemm_rows(r, "enc_proj", b->hook[1], D, ix.hi, Mhi, W("encoder.upsample_latent1.projection.weight"), dims[0], D,
                   nullptr, b->l1p, cpad(m, dims[0])));
  MD_TRY(deconv2(r, "enc_deconv", b->l1p, cpad(m, dims[0]), nullptr, hi, hi, W("encoder.upsample_latent1.upsample.0x1"),
                 cpad(m, dims[0]), dims[0], nullptr, b->enc1, cpad(m, dims[0]), 0, nullptr, 4, 3));
  // x0 (encoder.rs:153,425)
  MD_TRY(gemm_rows(r, "enc_proj", b->tok, D, ix.hi, Mhi, W("encoder.upsample0.projection.weight"), dims[1], D, nullptr,
                   b->x0p, cpad(m, dims[1])));
  MD_TRY(deconv2(r, "enc_deconv", b->x0p, cpad(m, dims[1]), nullptr, hi, hi, W("encoder.upsample0.upsample.0.weight"),
                 cpad(m, dims[1]), dims[1], nullptr, b->enc2, cpad(m, dims[1]), 0));
  // x1 (encoder.rs:154,426)
  MD_TRY(gemm_rows(r, "enc_proj", b->tok, D, ix.mid, Mmid, W("encoder.upsample1.projection.weight"), dims[2], D, nullptr,
                   b->x1p, cpad(m, dims[2])));
  MD_TRY(deconv2(r, "enc_deconv", b->x1p, cpad(m, dims[2]), nullptr, mid, mid, W("encoder.upsample1.upsample.0.weight"),
                 cpad(m, dims[2]), dims[2], nullptr, b->enc3, cpad(m, dims[2]), 0));
  // x2 + global image features -> cat -> fuse (encoder.rs:409-421)
  const int catld = 2 * cpad(m, dims[3]);
  MD_TRY(gemm_rows(r, "enc_proj", b->tok, D, ix.x2, Mlo, W("encoder.upsample2.projection.weight"), dims[3], D, nullptr,
                   b->x2p, cpad(m, dims[3])));
  MD_TRY(deconv2(r, "enc_deconv", b->x2p, cpad(m, dims[3]), nullptr, g, g, W("encoder.upsample2.upsample.0.weight"),
                 cpad(m, dims[3]), dims[3], nullptr, b->cat, catld, 0));
  MD_TRY(deconv2(r, "enc_deconv", b->tok, D, ix.img, g, g, W("encoder.upsample_lowres.weight"), D, dims[3],
                 P32(m, "encoder.upsample_lowres.bias"), b->cat, catld, cpad(m, dims[3])));
  MD_TRY(gemm_rows(r, "enc_fuse", b->cat, catld, nullptr, Mlo * 4, W("encoder.fuse_lowres.weight"), dims[3], catld,
                   P32(m, "encoder.fuse_lowres.bias"), b->enc4, cpad(m, dims[3])));
  if (m->taps_enabled) {
    MD_TRY(r.tap_nhwc("encoder_feature_0", b->enc0, F, 8 * hi, 8 * hi, cpad(m, F)));
    MD_TRY(r.tap_nhwc("encoder_feature_1", b->enc1, dims[0], 4 * hi, 4 * hi, cpad(m, dims[0])));
    MD_TRY(r.tap_nhwc("encoder_feature_2", b->enc2, dims[1], 2 * hi, 2 * hi, cpad(m, dims[1])));
    MD_TRY(r.tap_nhwc("encoder_feature_3", b->enc3, dims[2], 2 * mid, 2 * mid, cpad(m, dims[2])));
    MD_TRY(r.tap_nhwc("encoder_feature_4", b->enc4, dims[3], 2 * g, 2 * g, cpad(m, dims[3])));
  }
  return MD_OK;
}

static int run_decoder_head(Run& r, bool decoder_only = false) {
  md_model_s* m = r.m;
  md_model_s::Buffers* b = m->buf;
  const ModelCfg& c = m->cfg;
  const int F = c.F, Fp = cpad(m, F);
  const int* dims = c.pv.feat_dims;
  int hw[5];
  level_sizes(m, hw);
  const int ddims[5] = {F, dims[0], dims[1], dims[2], dims[3]};
  const void* enc[5] = {b->enc0, b->enc1, b->enc2, b->enc3, b->enc4};
  auto W = [&](const std::string& n) { return PK(m, n); };
  auto Bi = [&](const std::string& n) { return P32(m, n); };
  // ResidualBlock (decoder.rs:74-87): out = x + conv2(relu(conv1(relu(x)))) [+ extra]
  auto resblock = [&](const std::string& name, int l, const void* x, const void* xr, const void* extra, void* t,
                      void* out, void* out_relu) -> int {
    MD_TRY(conv3(r, "dec_conv3x3", xr, hw[l], hw[l], Fp, W(name + ".conv1.weight"), Bi(name + ".conv1.bias"), F, t, Fp,
                 ACT_RELU, nullptr, nullptr, nullptr));
    return conv3(r, "dec_conv3x3", t, hw[l], hw[l], Fp, W(name + ".conv2.weight"), Bi(name + ".conv2.bias"), F, out, Fp,
                 ACT_NONE, x, extra, out_relu);
  };
  const bool fused_c0 = PK(m, "head.outconv_conv0.weight") != nullptr && hw[0] >= 2;
  const void* feats = nullptr;
  for (int l = 4; l >= 0; --l) {
    const std::string f = "decoder.fusions." + std::to_string(l);
    const void *pj, *pjr;
    if (l == 0) {  // convs[0] is the identity (decoder.rs:155-165)
      pj = b->enc0;
      pjr = b->enc0r;
    } else {
      MD_TRY(conv3(r, "dec_conv3x3", enc[l], hw[l], hw[l], cpad(m, ddims[l]), W("decoder.convs." + std::to_string(l) + ".conv.weight"),
                   nullptr, F, b->proj[l], Fp, ACT_NONE, nullptr, nullptr, b->projr[l]));
      pj = b->proj[l];
      pjr = b->projr[l];
    }
    const void *x, *xr;
    if (l == 4) {  // top level: no skip input, resnet1 unused (decoder.rs:207-210)
      if (m->taps_enabled) MD_TRY(r.tap_nhwc("decoder_lowres_feature", pj, F, hw[4], hw[4], Fp));
      x = pj;
      xr = pjr;
    } else {
      MD_TRY(resblock(f + ".resnet1", l, pj, pjr, feats, b->dt[l], b->dx[l], b->dxr[l]));
      x = b->dx[l];
      xr = b->dxr[l];
    }
    MD_TRY(resblock(f + ".resnet2", l, x, xr, nullptr, b->dt[l], b->dy[l], nullptr));
    int ohw = hw[l];
    if (l != 0) {
      // deconv (no bias) then 1x1 out_conv (decoder.rs:124-141): one GEMM on the weight product packed at commit
      MD_TRY(deconv2(r, "dec_deconv_out", b->dy[l], Fp, nullptr, hw[l], hw[l], W(f + ".deconv_out_conv"), Fp, F,
                     Bi(f + ".out_conv.bias"), b->df[l], Fp, 0, nullptr, 2, 3));
      ohw = 2 * hw[l];
    } else if (!fused_c0 || m->taps_enabled) {  // level 0: the product path composes this 1x1 into head.conv0 (below);
                                                // its output exists only for the taps
      MD_TRY(gemm_rows(r, "dec_out_conv", b->dy[l], Fp, nullptr, (long)r.B * ohw * ohw, W(f + ".out_conv.weight"), F, Fp,
                       Bi(f + ".out_conv.bias"), b->df[l], Fp));
    }
    feats = b->df[l];
    if (m->taps_enabled) {
      const std::string tn = "decoder_fusion_" + std::to_string(l);
      MD_TRY(r.tap_nhwc(tn.c_str(), feats, F, ohw, ohw, Fp));
    }
  }
  if (m->taps_enabled) MD_TRY(r.tap_nhwc("decoder_feature", feats, F, hw[0], hw[0], Fp));
  if (decoder_only) return MD_OK;  // decoder_from_features (mod.rs:262-267)
  // depth head (mod.rs:105-112)
  const int F2 = F / 2, F2p = cpad(m, F2);
  if (fused_c0) {
    // out_conv 1x1 (+bias) -> conv0 3x3 (decoder.rs:137 -> mod.rs:105) as ONE 3x3 convolution on the last residual block's
    // output: the convolution adds the interior bias class, the border pixels get their class afterwards
    const float* bias9 = (const float*)PK(m, "head.outconv_conv0.bias");
    MD_TRY(conv3(r, "head_conv0", b->dy[0], hw[0], hw[0], Fp, W("head.outconv_conv0.weight"), bias9 + 4 * F2, F2, b->h0, F2p,
                 ACT_NONE, nullptr, nullptr, nullptr, 3));
    r.begin("head_conv0");
    MD_TRY(launch_border_bias_fix(b->h0, r.B, hw[0], hw[0], F2, F2p, bias9, m->prec, r.st));
    r.end();
  } else {
    MD_TRY(conv3(r, "head_conv0", feats, hw[0], hw[0], Fp, W("head.conv0.weight"), Bi("head.conv0.bias"), F2, b->h0, F2p,
                 ACT_NONE, nullptr, nullptr, nullptr));
  }
  if (m->taps_enabled) {
    // the deconv's output exists only as a debug tap: the product path below never materialises the 2x-resolution map
    MD_TRY(deconv2(r, "head_deconv_tap", b->h0, F2p, nullptr, hw[0], hw[0], W("head.deconv.weight"), F2p, F2,
                   Bi("head.deconv.bias"), b->h1, F2p, 0));
    MD_TRY(r.tap_nhwc("head_conv0", b->h0, F2, hw[0], hw[0], F2p));
    MD_TRY(r.tap_nhwc("head_deconv", b->h1, F2, 2 * hw[0], 2 * hw[0], F2p));
  }
  {
    // deconv k2s2 -> conv1 3x3 -> relu -> conv_out 1x1 -> relu (mod.rs:106-111) as ONE 3x3 convolution on conv0's output:
    // deconv and conv1 have nothing between them, so their product is packed at commit (add_pack_head_fused) as a 3x3
    // weight with 4 x 32 output columns, one 32-column group per output parity; the epilogue finishes the 1x1 tail.
    GemmParams p;
    p.N = 4 * 32; p.ngroups = 1; p.g_rows[0] = r.B * hw[0] * hw[0]; p.W[0] = W("head.deconv_conv1.weight");
    p.A = b->h0; p.cH = hw[0]; p.cW = hw[0]; p.zero_page = m->zero_page;
    split_conv_a(m, p, F2p, 3);
    p.epi = EPI_HEAD_UP2; p.bias[0] = (const float*)PK(m, "head.deconv_conv1.bias"); p.head_w = Bi("head.conv_out.weight");
    p.head_b = model_root(m)->head_b_host;
    p.out = b->canonical;
    r.begin("head_tail_fused");
    MD_TRY(launch_gemm(p, A_CONV3, m->prec, TILE_128x128, r.st));
    r.end();
  }
  MD_TRY(r.tap_f32("canonical_inverse_depth", b->canonical, r.B, 1, 2 * hw[0], 2 * hw[0]));
  return MD_OK;
}

// ---- debug entries on caller tensors: DepthPro::decoder_from_features / head_debug (mod.rs:262-307) ----
namespace {
struct DeviceScratch {  // per-call device memory of the debug entries (not a hot path)
  void* p = nullptr;
  ~DeviceScratch() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; MD_FAIL(MD_ERR_OOM, "hipMalloc of %zu bytes for a debug entry failed", bytes); }
    return MD_OK;
  }
};
struct TapsOn {  // the debug entries emit through the tap machinery whatever the model's tap switch says
  md_model_s* m; bool was;
  explicit TapsOn(md_model_s* mm) : m(mm), was(mm->taps_enabled) { m->taps_enabled = true; }
  ~TapsOn() { m->taps_enabled = was; }
};
// With the model's tap switch OFF, the fp32 tap buffers a debug entry creates (five [B,256,768,768] fusions + the head maps: ~3 GB at B = 1 of
// the default configuration) are released when the entry returns; buffers that existed before the call stay. Called behind the
// entry's stream synchronisation.
struct DebugTapScope {
  md_model_s* m; bool was; std::vector<std::string> before;
  explicit DebugTapScope(md_model_s* mm) : m(mm), was(mm->taps_enabled) {
    if (!was) for (auto& kv : m->taps) before.push_back(kv.first);
  }
  ~DebugTapScope() {
    if (was) return;
    for (auto it = m->taps.begin(); it != m->taps.end();) {
      if (std::find(before.begin(), before.end(), it->first) == before.end()) {
        if (it->second.dev) (void)hipFree(it->second.dev);
        it = m->taps.erase(it);
      } else {
        ++it;
      }
    }
  }
};
}  // namespace

static void decoder_level_shapes(md_model_s* m, int ddims[5], int hw[5]) {
  level_sizes(m, hw);
  const int* dims = m->cfg.pv.feat_dims;
  ddims[0] = m->cfg.F; ddims[1] = dims[0]; ddims[2] = dims[1]; ddims[3] = dims[2]; ddims[4] = dims[3];
}

bool model_decoder_query(md_model_t m, const std::string& key, int64_t* out) {
  if (!m || m->kind != 0) return false;
  int ddims[5], hw[5];
  decoder_level_shapes(m, ddims, hw);
  if (key == "decoder_levels") { *out = 5; return true; }
  if (key == "decoder_features") { *out = m->cfg.F; return true; }
  for (int l = 0; l < 5; ++l) {
    if (key == "decoder_level" + std::to_string(l) + "_channels") { *out = ddims[l]; return true; }
    if (key == "decoder_level" + std::to_string(l) + "_size") { *out = hw[l]; return true; }
  }
  return false;
}

// caller NCHW fp32 (host or device) -> NHWC T rows of `ld` logical channels (+ an optional relu'd copy)
static int stage_nchw_feature(md_model_s* m, hipStream_t st, const md_nchw_view& v, int B, int in_kind, float* stage, void* dst, void* dst_relu,
                              long ld) {
  const size_t n = (size_t)B * v.channels * v.height * v.width;
  const float* src = v.data;
  if (in_kind == MD_MEM_HOST) {
    MD_HIP(hipMemcpyAsync(stage, v.data, n * 4, hipMemcpyHostToDevice, st));
    src = stage;
  }
  MD_TRY(launch_nchw_to_nhwc(src, B, v.channels, v.height, v.width, dst, m->prec, 0, st, ld));
  if (dst_relu) MD_TRY(launch_nchw_to_nhwc(src, B, v.channels, v.height, v.width, dst_relu, m->prec, 1, st, ld));
  return MD_OK;
}

static int copy_tap_out(md_model_s* m, const char* name, float* out, int out_kind, hipStream_t st) {
  if (!out) return MD_OK;
  auto it = m->taps.find(name);
  if (it == m->taps.end() || !it->second.dev) MD_FAIL(MD_ERR_HIP, "internal: tensor `%s` was not produced", name);
  if (out_kind == MD_MEM_DEVICE) {
    MD_HIP(hipMemcpyAsync(out, it->second.dev, it->second.count * 4, hipMemcpyDeviceToDevice, st));
  } else {
    MD_HIP(hipStreamSynchronize(st));
    MD_HIP(hipMemcpy(out, it->second.dev, it->second.count * 4, hipMemcpyDeviceToHost));
  }
  return MD_OK;
}

static int debug_entry_checks(md_model_t m, int B, int in_kind, int out_kind) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  if (m->kind != 0) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth Pro model");
  if (!model_root(m)->committed) MD_FAIL(MD_ERR_INVALID_ARG, "weights were modified; call md_model_commit_weights first");
  if (B <= 0 || B > m->cfg.max_batch) MD_FAIL(MD_ERR_SHAPE, "batch %d outside 1 .. max_batch %d", B, m->cfg.max_batch);
  if ((in_kind != MD_MEM_HOST && in_kind != MD_MEM_DEVICE) || (out_kind != MD_MEM_HOST && out_kind != MD_MEM_DEVICE))
    MD_FAIL(MD_ERR_INVALID_ARG, "memory kind %d / %d", in_kind, out_kind);
  return MD_OK;
}

int model_decoder_from_features(md_model_t m, const md_nchw_view* features, int levels, int B, int in_kind, float* out_features,
                                float* out_lowres, float* const* out_fusions, int out_kind, hipStream_t stream) {
  MD_TRY(debug_entry_checks(m, B, in_kind, out_kind));
  if (!features) MD_FAIL(MD_ERR_INVALID_ARG, "features pointer is null");
  if (levels != 5) MD_FAIL(MD_ERR_LEVELS, "Got encoder output levels = %d, expected 5.", levels);  // decoder.rs:200-205
  int ddims[5], hw[5];
  decoder_level_shapes(m, ddims, hw);
  size_t max_elems = 0;
  for (int l = 0; l < 5; ++l) {
    const md_nchw_view& v = features[l];
    if (!v.data) MD_FAIL(MD_ERR_INVALID_ARG, "features[%d].data is null", l);
    if (v.channels != ddims[l] || v.height != hw[l] || v.width != hw[l])
      MD_FAIL(MD_ERR_SHAPE, "features[%d] is [B,%d,%d,%d]; this model's decoder takes [B,%d,%d,%d]", l, v.channels, v.height, v.width, ddims[l],
              hw[l], hw[l]);
    max_elems = std::max(max_elems, (size_t)B * v.channels * v.height * v.width);
  }
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t st = stream ? stream : (m->own_stream ? m->own_stream : m->dev->stream);
  md_model_s::Buffers* b = m->buf;
  DeviceScratch stage;
  if (in_kind == MD_MEM_HOST) MD_TRY(stage.alloc(max_elems * 4));
  void* enc[5] = {b->enc0, b->enc1, b->enc2, b->enc3, b->enc4};
  for (int l = 0; l < 5; ++l)  // level 0 also feeds resnet1 through its relu'd copy (convs[0] is the identity, decoder.rs:155-165)
    MD_TRY(stage_nchw_feature(m, st, features[l], B, in_kind, (float*)stage.p, enc[l], l == 0 ? b->enc0r : nullptr, cpad(m, ddims[l])));
  Run r{m, st, B};
  DebugTapScope tap_scope(m);
  {
    TapsOn taps(m);
    MD_TRY(run_decoder_head(r, true));
  }
  MD_TRY(copy_tap_out(m, "decoder_feature", out_features, out_kind, st));
  MD_TRY(copy_tap_out(m, "decoder_lowres_feature", out_lowres, out_kind, st));
  if (out_fusions)
    for (int l = 0; l < 5; ++l) MD_TRY(copy_tap_out(m, ("decoder_fusion_" + std::to_string(l)).c_str(), out_fusions[l], out_kind, st));
  MD_HIP(hipStreamSynchronize(st));  // the per-call staging is freed on return
  return MD_OK;
}

int model_head_debug(md_model_t m, const md_nchw_view* feature, int B, int in_kind, const md_head_debug* out, int out_kind,
                     hipStream_t stream) {
  MD_TRY(debug_entry_checks(m, B, in_kind, out_kind));
  if (!feature || !feature->data || !out) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  int ddims[5], hw[5];
  decoder_level_shapes(m, ddims, hw);
  const int F = m->cfg.F, Fp = cpad(m, F), F2 = F / 2, F2p = cpad(m, F2), C1 = 32, C1p = cpad(m, C1), s0 = hw[0];
  if (feature->channels != F || feature->height != s0 || feature->width != s0)
    MD_FAIL(MD_ERR_SHAPE, "feature is [B,%d,%d,%d]; this model's head takes [B,%d,%d,%d]", feature->channels, feature->height, feature->width, F, s0, s0);
  if (!PK(m, "head.conv1.weight")) MD_FAIL(MD_ERR_UNSUPPORTED, "head_debug: the un-fused conv1 operand is not packed");
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t st = stream ? stream : (m->own_stream ? m->own_stream : m->dev->stream);
  md_model_s::Buffers* b = m->buf;
  const size_t px1 = (size_t)B * 4 * s0 * s0;
  const size_t map_bytes = align_up(px1 * C1p * m->esz * m->xm + 256, 256);
  DeviceScratch stage, maps, tails;
  if (in_kind == MD_MEM_HOST) MD_TRY(stage.alloc((size_t)B * F * s0 * s0 * 4));
  MD_TRY(maps.alloc(2 * map_bytes));
  MD_TRY(tails.alloc(2 * px1 * 4));
  void* c1 = maps.p;
  void* c1r = (char*)maps.p + map_bytes;
  float* pre = (float*)tails.p;
  float* can = pre + px1;
  auto W = [&](const char* n) { return PK(m, n); };
  auto Bi = [&](const char* n) { return P32(m, n); };
  MD_TRY(stage_nchw_feature(m, st, *feature, B, in_kind, (float*)stage.p, b->df[0], nullptr, Fp));
  Run r{m, st, B};
  DebugTapScope tap_scope(m);
  TapsOn taps(m);
  MD_TRY(conv3(r, "head_conv0", b->df[0], s0, s0, Fp, W("head.conv0.weight"), Bi("head.conv0.bias"), F2, b->h0, F2p, ACT_NONE, nullptr,
               nullptr, nullptr));
  MD_TRY(deconv2(r, "head_deconv_tap", b->h0, F2p, nullptr, s0, s0, W("head.deconv.weight"), F2p, F2, Bi("head.deconv.bias"), b->h1, F2p, 0));
  MD_TRY(conv3(r, "head_conv1_debug", b->h1, 2 * s0, 2 * s0, F2p, W("head.conv1.weight"), Bi("head.conv1.bias"), C1, c1, C1p, ACT_NONE,
               nullptr, nullptr, c1r));
  MD_TRY(launch_head_tail_debug(c1r, C1p, (long)px1, C1, Bi("head.conv_out.weight"), Bi("head.conv_out.bias"), pre, can, m->prec, st));
  MD_TRY(r.tap_nhwc("head_conv0", b->h0, F2, s0, s0, F2p));
  MD_TRY(r.tap_nhwc("head_deconv", b->h1, F2, 2 * s0, 2 * s0, F2p));
  MD_TRY(r.tap_nhwc("head_conv1", c1, C1, 2 * s0, 2 * s0, C1p));
  MD_TRY(r.tap_nhwc("head_relu", c1r, C1, 2 * s0, 2 * s0, C1p));
  MD_TRY(r.tap_f32("head_pre_out", pre, B, 1, 2 * s0, 2 * s0));
  MD_TRY(r.tap_f32("head_canonical", can, B, 1, 2 * s0, 2 * s0));
  MD_TRY(copy_tap_out(m, "head_conv0", out->conv0, out_kind, st));
  MD_TRY(copy_tap_out(m, "head_deconv", out->deconv, out_kind, st));
  MD_TRY(copy_tap_out(m, "head_conv1", out->conv1, out_kind, st));
  MD_TRY(copy_tap_out(m, "head_relu", out->relu, out_kind, st));
  MD_TRY(copy_tap_out(m, "head_pre_out", out->pre_out, out_kind, st));
  MD_TRY(copy_tap_out(m, "head_canonical", out->canonical, out_kind, st));
  MD_HIP(hipStreamSynchronize(st));  // the per-call scratch is freed on return
  return MD_OK;
}

// NHWC f32 bilinear (ensure_min_spatial, fov.rs:238-246) -- tiny maps only, so go through NCHW
static int fov_min_spatial(Run& r, float** cur, int* h, int C, int kmin, float* scratch_a, float* scratch_b) {
  if (*h >= kmin) return MD_OK;
  md_model_s* m = r.m;
  const int B = r.B, oh = kmin;
  // NHWC f32 -> NCHW f32
  r.begin("fov_resize");
  MD_TRY(launch_nhwc_to_nchw(*cur, B, C, *h, *h, C, 0, scratch_a, MD_PREC_F32, r.st));
  r.end();
  r.begin("fov_resize");
  MD_TRY(launch_resize_bilinear(scratch_a, B * C, *h, *h, scratch_b, oh, oh, m->cfg.interpolation, 0, r.st));
  r.end();
  r.begin("fov_resize");
  MD_TRY(launch_nchw_to_nhwc(scratch_b, B, C, oh, oh, scratch_a, MD_PREC_F32, 0, r.st));
  r.end();
  *cur = scratch_a;
  *h = oh;
  return MD_OK;
}

static int run_fov(Run& r, const md_model_s::IndexSet& ix) {
  md_model_s* m = r.m;
  md_model_s::Buffers* b = m->buf;
  const ModelCfg& c = m->cfg;
  const int B = r.B, F = c.F, g = m->g;
  int hw[5];
  level_sizes(m, hw);
  const void* lowres = b->proj[4];  // convs[4](enc[4]) (decoder.rs:207-208)
  const int Fp = cpad(m, F);
  auto Wd = [&](const std::string& n) { return (const float*)PK(m, n); };
  auto Bi = [&](const std::string& n) { return P32(m, n); };
  if (Fp != F) MD_FAIL(MD_ERR_UNSUPPORTED, "fov: decoder_features must be a multiple of the MFMA k-tile");
  float* stage[4] = {b->fv0, b->fv1, b->fv2, b->fv3};
  float* sa = b->fvr;
  float* sb = b->fvr + (size_t)B * 36 * F;
  if (c.has_fov_vit) {
    r.begin("fov_head");
    // fov.rs:178-227: downsample(lowres) + Linear(tokens) -> head convs
    int h = (hw[4] + 2 - 3) / 2 + 1;
    if (PK(m, "fov.downsample_blocks.0.conv.gemm") && (F / 2) % 4 == 0) {
      GemmParams p;
      p.N = F / 2; p.ngroups = 1; p.g_rows[0] = B * h * h;
      p.W[0] = PK(m, "fov.downsample_blocks.0.conv.gemm"); p.bias[0] = Bi("fov.downsample_blocks.0.conv.bias");
      p.A = lowres; p.cH = hw[4]; p.cW = hw[4]; p.cOH = h; p.cOW = h; p.cstride = 2; p.zero_page = m->zero_page;
      split_conv_a(m, p, Fp, 0);
      p.epi = EPI_STORE; p.act = ACT_RELU; p.out_f32 = 1; p.out = stage[0]; p.ldo = F / 2;
      MD_TRY(launch_gemm(p, A_CONV3, m->prec, TILE_AUTO, r.st));
    } else {
      MD_TRY(launch_conv_direct(lowres, m->prec, nullptr, B, hw[4], hw[4], F, Wd("fov.downsample_blocks.0.conv.weight"),
                                Bi("fov.downsample_blocks.0.conv.bias"), F / 2, 3, 2, 1, 1, stage[0], r.st));
    }
    if (h != g) MD_FAIL(MD_ERR_UNSUPPORTED, "fov: downsampled lowres %d does not match the token grid %d", h, g);
    r.end();
    MD_TRY(gemm_rows(r, "fov_proj", b->tok, c.pv.D, ix.fov, (long)B * m->P, PK(m, "fov.encoder_proj.weight"), F / 2, c.pv.D,
                     Bi("fov.encoder_proj.bias"), b->fovproj, F / 2, 1));
    float* cur = stage[0];
    int ch = F / 2;
    const float* add = b->fovproj;
    const char* names[3] = {"fov.head_blocks.0.conv", "fov.head_blocks.1.conv", "fov.head_blocks.2.conv"};
    const int couts[3] = {F / 4, F / 8, 1}, ks[3] = {3, 3, 6}, strides[3] = {2, 2, 1}, pads[3] = {1, 1, 0},
              relus[3] = {1, 1, 0};
    for (int i = 0; i < 3; ++i) {
      if (h < ks[i]) {
        if (add) MD_FAIL(MD_ERR_UNSUPPORTED, "fov: resize before the fused add is not supported");
        MD_TRY(fov_min_spatial(r, &cur, &h, ch, ks[i], sa, sb));
      }
      float* out = i == 2 ? b->fov_deg : stage[i + 1];
      r.begin("fov_head");
      MD_TRY(launch_conv_direct(cur, MD_PREC_F32, add, B, h, h, ch, Wd(std::string(names[i]) + ".weight"),
                                Bi(std::string(names[i]) + ".bias"), couts[i], ks[i], strides[i], pads[i], relus[i], out,
                                r.st));
      r.end();
      add = nullptr;
      h = (h + 2 * pads[i] - ks[i]) / strides[i] + 1;
      ch = couts[i];
      cur = out;
    }
    if (h != 1) MD_FAIL(MD_ERR_UNSUPPORTED, "fov head ends at %dx%d, expected 1x1", h, h);
  } else {
    // fov.rs:118-155: four head blocks straight on the lowres feature
    const char* names[4] = {"fov.head_blocks.0.conv", "fov.head_blocks.1.conv", "fov.head_blocks.2.conv",
                            "fov.head_blocks.3.conv"};
    const int couts[4] = {F / 2, F / 4, F / 8, 1}, ks[4] = {3, 3, 3, 6}, strides[4] = {2, 2, 2, 1}, pads[4] = {1, 1, 1, 0},
              relus[4] = {1, 1, 1, 0};
    const void* cur = lowres;
    int cur_prec = m->prec, h = hw[4], ch = F;
    for (int i = 0; i < 4; ++i) {
      if (h < ks[i]) {
        if (cur_prec != MD_PREC_F32) MD_FAIL(MD_ERR_UNSUPPORTED, "fov: lowres smaller than the first kernel");
        float* cf = (float*)cur;
        MD_TRY(fov_min_spatial(r, &cf, &h, ch, ks[i], sa, sb));
        cur = cf;
      }
      float* out = i == 3 ? b->fov_deg : stage[i];
      r.begin("fov_head");
      MD_TRY(launch_conv_direct(cur, cur_prec, nullptr, B, h, h, ch, Wd(std::string(names[i]) + ".weight"),
                                Bi(std::string(names[i]) + ".bias"), couts[i], ks[i], strides[i], pads[i], relus[i], out,
                                r.st));
      r.end();
      h = (h + 2 * pads[i] - ks[i]) / strides[i] + 1;
      ch = couts[i];
      cur = out;
      cur_prec = MD_PREC_F32;
    }
    if (h != 1) MD_FAIL(MD_ERR_UNSUPPORTED, "fov head ends at %dx%d, expected 1x1", h, h);
  }
  MD_TRY(r.tap_f32("fov_deg", b->fov_deg, r.B, 1, 1, 1));
  return MD_OK;
}

static int model_infer_eager(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, float* focal,
                             float* fovx, float* fovy, int out_kind, hipStream_t stream, const uint8_t* rgb, size_t rgb_len,
                             const ShardPlan* sp = nullptr);

// Grow-only staging: (re)allocates only when `need` exceeds the capacity. hipFree is a device-wide synchronisation and
// hipMalloc takes the allocator lock, so a caller that feeds host pointers or a fixed non-native size (the reference's
// `infer_from_rgb` path, src/inference.rs:128-137; the auto-resize of mod.rs:312-325) must not pay either per call.
// md_model_query("allocs") counts what the infer calls of a model have allocated.
static int ensure_device(md_model_s* m, void** p, size_t* cap, size_t need) {
  if (*cap >= need && *p) return MD_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  MD_HIP(hipMalloc(p, need));
  *cap = need;
  m->alloc_count += 1;
  return MD_OK;
}
static int ensure_pinned(md_model_s* m, void** p, size_t* cap, size_t need) {
  if (*cap >= need && *p) return MD_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  MD_HIP(hipHostMalloc(p, need, hipHostMallocDefault));
  *cap = need;
  m->alloc_count += 1;
  return MD_OK;
}
// pageable caller memory -> pinned bounce buffer -> device, asynchronously on `st` (the bounce buffer is reused by the next
// call, which first waits for this stream's work: one in-flight infer per model, see the threading rule in mi_depth.h)
static int stage_host_to_device(md_model_s* m, void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
  md_model_s::Buffers* b = m->buf;
  if (b->pin_in_cap < bytes) MD_HIP(hipStreamSynchronize(st));  // nothing may still read the buffer being replaced
  MD_TRY(ensure_pinned(m, &b->pin_in, &b->pin_in_cap, bytes));
  MD_HIP(hipStreamSynchronize(st));  // the previous call's copy out of the bounce buffer has finished
  memcpy(b->pin_in, src_host, bytes);
  MD_HIP(hipMemcpyAsync(dst_dev, b->pin_in, bytes, hipMemcpyHostToDevice, st));
  return MD_OK;
}

int model_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, float* focal,
                float* fovx, float* fovy, int out_kind, hipStream_t stream, const uint8_t* rgb, size_t rgb_len) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  auto body = [&]() { return model_infer_eager(m, nchw, B, H, W, in_kind, depth, focal, fovx, fovy, out_kind, stream, rgb, rgb_len); };
  if (!m->graph_enabled) return body();
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t st = stream ? stream : (m->own_stream ? m->own_stream : m->dev->stream);
  const bool eligible = nchw && !rgb && in_kind == MD_MEM_DEVICE && out_kind == MD_MEM_DEVICE && model_root(m)->committed && B > 0 &&
                        B <= m->cfg.max_batch && H == m->S && W == m->S;
  // the commit generation of the weights is part of the key: a graph bakes by-value launch parameters (the head's output
  // bias, the split-half term count), and a fork's graphs cannot be reached from the root's commit
  const unsigned gen = model_root(m)->commit_gen;
  if (m->graphs_gen != gen) {  // a fork's graphs of an older commit can never be replayed again (the root clears its own in model_commit)
    for (auto& kv : m->graphs)
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    m->graphs.clear();
    m->graphs_gen = gen;
  }
  const std::vector<uintptr_t> key = {(uintptr_t)st, (uintptr_t)B, (uintptr_t)H, (uintptr_t)W, (uintptr_t)nchw, (uintptr_t)depth,
                                      (uintptr_t)focal, (uintptr_t)fovx, (uintptr_t)fovy, (uintptr_t)gen};
  return run_with_graph(m, st, key, eligible, body);
}

int model_stage_input(md_model_t m, const float* nchw, size_t elems, int in_kind, hipStream_t stream, float** dev) {
  if (!m || !dev || elems == 0) MD_FAIL(MD_ERR_INVALID_ARG, "model_stage_input: null argument");
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t st = stream ? stream : (m->own_stream ? m->own_stream : m->dev->stream);
  md_model_s::Buffers* b = m->buf;
  if (b->xraw_cap < elems * 4) MD_HIP(hipStreamSynchronize(st));  // nothing may still read the buffer being replaced
  MD_TRY(ensure_device(m, (void**)&b->xraw, &b->xraw_cap, elems * 4));
  if (nchw) {
    if (in_kind == MD_MEM_HOST) MD_TRY(stage_host_to_device(m, b->xraw, nchw, elems * 4, st));
    else MD_HIP(hipMemcpyAsync(b->xraw, nchw, elems * 4, hipMemcpyDeviceToDevice, st));
  }
  *dev = b->xraw;
  return MD_OK;
}

int model_infer_sharded(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, float* focal,
                        float* fovx, float* fovy, int out_kind, hipStream_t stream, const ShardPlan& sp) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  if (sp.parts < 1 || sp.part < -1 || sp.part >= sp.parts || sp.root < 0 || sp.root >= sp.parts)
    MD_FAIL(MD_ERR_INVALID_ARG, "tile-parallel plan: part %d of %d, root %d", sp.part, sp.parts, sp.root);
  if (sp.part >= 0 && sp.parts > 1 && !sp.exchange) MD_FAIL(MD_ERR_INVALID_ARG, "tile-parallel plan: no exchange function");
  return model_infer_eager(m, nchw, B, H, W, in_kind, depth, focal, fovx, fovy, out_kind, stream, nullptr, 0, &sp);
}

static int model_infer_eager(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, float* focal,
                             float* fovx, float* fovy, int out_kind, hipStream_t stream, const uint8_t* rgb, size_t rgb_len,
                             const ShardPlan* sp) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  if (!model_root(m)->committed) MD_FAIL(MD_ERR_INVALID_ARG, "weights were modified; call md_model_commit_weights first");
  if (!nchw && !rgb) MD_FAIL(MD_ERR_INVALID_ARG, "input pointer is null");
  if (B <= 0 || H <= 0 || W <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid input shape [%d,3,%d,%d]", B, H, W);
  if (B > m->cfg.max_batch) MD_FAIL(MD_ERR_SHAPE, "batch %d exceeds max_batch %d", B, m->cfg.max_batch);
  if (!m->cfg.use_fov_head) MD_FAIL(MD_ERR_NO_FOV, "FOV head required for focal length");  // mod.rs:329
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t st = stream ? stream : (m->own_stream ? m->own_stream : m->dev->stream);
  md_model_s::Buffers* b = m->buf;
  Run r{m, st, B};
  const int S = m->S;
  const size_t in_elems = (size_t)B * 3 * H * W;
  const bool resize_needed = H != S || W != S;  // mod.rs:317
  const float* x_dev = nullptr;
  // ---- stage the input ----
  if (rgb) {
    if (rgb_len != (size_t)W * H * 3) MD_FAIL(MD_ERR_SHAPE, "expected %zu RGB bytes for %dx%d, got %zu", (size_t)W * H * 3, W, H, rgb_len);
    const uint8_t* rgb_dev = rgb;
    if (in_kind == MD_MEM_HOST) {
      MD_TRY(ensure_device(m, (void**)&b->rgb, &b->rgb_cap, rgb_len));
      MD_TRY(stage_host_to_device(m, b->rgb, rgb, rgb_len, st));
      rgb_dev = b->rgb;
    }
    float* dst = b->xin;
    if (resize_needed) {
      MD_TRY(ensure_device(m, (void**)&b->xraw, &b->xraw_cap, in_elems * 4));
      dst = b->xraw;
    }
    r.begin("rgb_to_input");
    MD_TRY(launch_rgb_to_input(rgb_dev, W, H, dst, st));
    r.end();
    x_dev = dst;
  } else if (in_kind == MD_MEM_HOST) {
    float* dst = b->xin;
    if (resize_needed) {
      MD_TRY(ensure_device(m, (void**)&b->xraw, &b->xraw_cap, in_elems * 4));
      dst = b->xraw;
    }
    MD_TRY(stage_host_to_device(m, dst, nchw, in_elems * 4, st));
    x_dev = dst;
  } else {
    x_dev = nchw;
  }
  if (resize_needed) {
    r.begin("resize_in");
    MD_TRY(launch_resize_bilinear(x_dev, B * 3, H, W, b->xin, S, S, m->cfg.interpolation, 0, st));
    r.end();
    x_dev = b->xin;
  }
  // ---- encoder ----
  const int n0 = m->steps0 * m->steps0 * B, n1 = m->steps1 * m->steps1 * B;
  const int nseq_p = n0 + n1 + B;
  const int nseq = nseq_p + B * (m->ngroups - 1);
  PyramidGeom pg{B, S, m->win, m->cfg.pv.ps, m->steps0, m->stride0, m->steps1, m->stride1, m->cfg.interpolation};
  r.begin("pyramid_patchify");
  MD_TRY(launch_pyramid_patchify(x_dev, pg, b->patches, m->prec, st));
  r.end();
  md_model_s::IndexSet ix;
  MD_TRY(get_index_set(m, B, &ix));
  // diagnostic timing of the windowed pass (sp->part == -1); destroyed on EVERY way out of this call, early error returns included
  struct EventSet {
    hipEvent_t e[66] = {};
    ~EventSet() {
      for (hipEvent_t x : e)
        if (x) (void)hipEventDestroy(x);
    }
  } ev_set;
  hipEvent_t* ev_w = ev_set.e;
  const bool time_windows = sp && sp->part < 0 && (sp->window_ms || sp->tail_ms) && sp->parts <= 64;
  if (!sp) {
    MD_TRY(run_vit(r, nseq_p, nseq, 0, nseq));
  } else {
    auto lo_of = [&](int p) { return (int)((long)nseq * p / sp->parts); };
    if (sp->part < 0) {  // every window in turn on this device: the same launches a rank of each part would issue
      for (int p = 0; p < sp->parts; ++p) {
        if (time_windows) { MD_HIP(hipEventCreate(&ev_w[p])); MD_HIP(hipEventRecord(ev_w[p], st)); }
        MD_TRY(run_vit(r, nseq_p, nseq, lo_of(p), lo_of(p + 1)));
      }
      if (time_windows) { MD_HIP(hipEventCreate(&ev_w[sp->parts])); MD_HIP(hipEventRecord(ev_w[sp->parts], st)); }
    } else {
      MD_TRY(run_vit(r, nseq_p, nseq, lo_of(sp->part), lo_of(sp->part + 1)));
      if (sp->parts > 1) {
        const size_t rowb = (size_t)m->SS * m->cfg.pv.D * m->esz * m->xm;  // bytes of one sequence of a [rows, D] T tensor
        std::vector<ShardSegment> segs((size_t)sp->parts * 3);
        for (int p = 0; p < sp->parts; ++p) {
          const int lo = lo_of(p), hi = lo_of(p + 1), hlo = std::min(lo, n0), hhi = std::min(hi, n0);
          segs[3 * p + 0] = {(char*)b->tok + (size_t)lo * rowb, (size_t)(hi - lo) * rowb};
          segs[3 * p + 1] = {(char*)b->hook[0] + (size_t)hlo * rowb, (size_t)(hhi - hlo) * rowb};
          segs[3 * p + 2] = {(char*)b->hook[1] + (size_t)hlo * rowb, (size_t)(hhi - hlo) * rowb};
        }
        MD_TRY(sp->exchange(sp->ctx, sp->parts, (const ShardSegment(*)[3])segs.data(), st));
      }
      if (sp->part != sp->root) return MD_OK;  // this rank's share ends here: the root owns the rest and the outputs
    }
  }
  auto finish_timing = [&]() -> int {
    if (!time_windows) return MD_OK;
    const int np = sp->parts;
    MD_HIP(hipEventCreate(&ev_w[np + 1]));
    MD_HIP(hipEventRecord(ev_w[np + 1], st));
    MD_HIP(hipEventSynchronize(ev_w[np + 1]));
    for (int p = 0; p < np; ++p)
      if (sp->window_ms) MD_HIP(hipEventElapsedTime(&sp->window_ms[p], ev_w[p], ev_w[p + 1]));
    if (sp->tail_ms) MD_HIP(hipEventElapsedTime(sp->tail_ms, ev_w[np], ev_w[np + 1]));
    return MD_OK;
  };
  MD_TRY(run_encoder_tail(r, ix));
  MD_TRY(run_decoder_head(r));
  MD_TRY(run_fov(r, ix));
  // ---- tail (mod.rs:330-363) ----
  r.begin("fov_post");
  MD_TRY(launch_fov_post(b->fov_deg, B, H, W, b->focal, b->fovy, b->ratio, st));
  r.end();
  const size_t out_elems = (size_t)B * H * W;
  float* depth_dev = nullptr;
  if (depth) {
    if (out_kind == MD_MEM_DEVICE) {
      depth_dev = depth;
    } else {
      size_t cap = b->depth_stage_elems * 4;
      MD_TRY(ensure_device(m, (void**)&b->depth_stage, &cap, out_elems * 4));
      b->depth_stage_elems = cap / 4;
      depth_dev = b->depth_stage;
    }
    r.begin("depth_post");
    if (!resize_needed) {
      MD_TRY(launch_depth_post(b->canonical, b->ratio, B, (long)S * S, depth_dev, 1, st));
      r.end();
    } else {
      MD_TRY(launch_depth_post(b->canonical, b->ratio, B, (long)S * S, b->inv, 0, st));
      r.end();
      r.begin("resize_out");
      MD_TRY(launch_resize_bilinear(b->inv, B, S, S, depth_dev, H, W, m->cfg.interpolation, 1, st));
      r.end();
    }
  }
  if (out_kind == MD_MEM_HOST) {
    // host outputs: one pinned bounce buffer [depth | focal | fovx | fovy], asynchronous device -> pinned copies, ONE stream
    // synchronisation, then plain memcpy into the caller's (pageable) memory
    const size_t dbytes = depth ? out_elems * 4 : 0, need = dbytes + 3 * (size_t)B * 4;
    MD_TRY(ensure_pinned(m, &b->pin_out, &b->pin_out_cap, need));
    char* ph = (char*)b->pin_out;
    if (depth) MD_HIP(hipMemcpyAsync(ph, depth_dev, dbytes, hipMemcpyDeviceToHost, st));
    const float* srcs[3] = {b->focal, b->fov_deg, b->fovy};
    float* dsts[3] = {focal, fovx, fovy};
    for (int i = 0; i < 3; ++i)
      if (dsts[i]) MD_HIP(hipMemcpyAsync(ph + dbytes + (size_t)i * B * 4, srcs[i], (size_t)B * 4, hipMemcpyDeviceToHost, st));
    MD_HIP(hipStreamSynchronize(st));
    if (depth) memcpy(depth, ph, dbytes);
    for (int i = 0; i < 3; ++i)
      if (dsts[i]) memcpy(dsts[i], ph + dbytes + (size_t)i * B * 4, (size_t)B * 4);
    return finish_timing();
  }
  auto copy_out = [&](float* dst, const float* src, size_t n) -> int {
    if (!dst) return MD_OK;
    MD_HIP(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, st));
    return MD_OK;
  };
  MD_TRY(copy_out(focal, b->focal, B));
  MD_TRY(copy_out(fovx, b->fov_deg, B));
  MD_TRY(copy_out(fovy, b->fovy, B));
  return finish_timing();
}

}  // namespace md
