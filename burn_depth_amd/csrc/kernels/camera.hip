// Depth-Anything-v3 camera ENCODER (`CameraEncoder::forward`, depth_anything3/camera.rs:89-110), one launch.
//
// The whole encoder works on V view tokens of width D per image (V is 1..16, D = 384 for `small`): pose encoding ->
// PoseBranch (fc1 9 -> D/2, erf-GELU, fc2 -> D) -> token_norm -> `depth` pre-norm transformer blocks over the V tokens ->
// trunk_norm -> mean over views. That is ~7 MFLOP per token against ~28 MB of fp32 weights: the cost is reading the weights
// once and a dependent chain of ~40 tiny steps. Composed from separate launches it would be ~40 launches of ~4.5 us inside a
// model whose whole step is ~2 ms; here ONE workgroup per image walks the chain with __syncthreads between the steps, every
// wave owning whole output columns of the current linear layer (lanes stride the reduction dimension: coalesced weight rows,
// one butterfly per column), activations in a small global scratch that stays in L2. fp32 throughout, in every precision mode
// of the engine: the result is one token of the fp32 residual stream.
#include <hip/hip_runtime.h>

#include "camera_math.h"
#include "ops.h"

namespace md {

namespace {

constexpr int kMaxViews = MD_CAM_MAX_VIEWS;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// camera.rs:516-536
__device__ __forceinline__ float atan_poly(float v) { return 0.78539816339744830962f * v - v * (v - 1.0f) * (0.2447f + 0.0663f * v); }
__device__ __forceinline__ float approx_atan_positive(float x) {
  const float small = atan_poly(x), large = 1.57079632679489661923f - atan_poly(1.0f / fmaxf(x, 1e-6f));
  const float mk = x <= 1.0f ? 1.0f : 0.0f;
  return small * mk + large * (1.0f - mk);
}

// extri_intri_to_pose_encoding for one view (camera.rs:236-279, 418-514): e = world-to-camera [3][4], k = intrinsics [3][3]
__device__ void pose_encode(const float* __restrict__ e, const float* __restrict__ k, float half_h, float half_w, float* __restrict__ out) {
  // camera-to-world rotation = R^T; translation = -(R^T t)
  float m[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m[i][j] = e[j * 4 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = -(m[i][0] * e[3] + m[i][1] * e[7] + m[i][2] * e[11]);
  const float trace = m[0][0] + m[1][1] + m[2][2], eps = 1e-6f;
  float s = sqrtf(fmaxf(trace + 1.0f, 1e-6f)) * 2.0f;
  const float qt[4] = {(m[2][1] - m[1][2]) / s, (m[0][2] - m[2][0]) / s, (m[1][0] - m[0][1]) / s, 0.25f * s};
  s = sqrtf(fmaxf(1.0f + m[0][0] - m[1][1] - m[2][2], 1e-6f)) * 2.0f;
  const float qx[4] = {0.25f * s, (m[0][1] + m[1][0]) / (s + eps), (m[0][2] + m[2][0]) / (s + eps), (m[2][1] - m[1][2]) / (s + eps)};
  s = sqrtf(fmaxf(1.0f + m[1][1] - m[0][0] - m[2][2], 1e-6f)) * 2.0f;
  const float qy[4] = {(m[0][1] + m[1][0]) / (s + eps), 0.25f * s, (m[1][2] + m[2][1]) / (s + eps), (m[0][2] - m[2][0]) / (s + eps)};
  s = sqrtf(fmaxf(1.0f + m[2][2] - m[0][0] - m[1][1], 1e-6f)) * 2.0f;
  const float qz[4] = {(m[0][2] + m[2][0]) / (s + eps), (m[1][2] + m[2][1]) / (s + eps), 0.25f * s, (m[1][0] - m[0][1]) / (s + eps)};
  const float mt = trace > 0.0f ? 1.0f : 0.0f;
  const float mx = (1.0f - mt) * (m[0][0] > m[1][1] ? 1.0f : 0.0f) * (m[0][0] > m[2][2] ? 1.0f : 0.0f);
  const float my = (1.0f - mt - mx) * (m[1][1] > m[2][2] ? 1.0f : 0.0f);
  const float mz = 1.0f - mt - mx - my;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[3 + i] = qt[i] * mt + qx[i] * mx + qy[i] * my + qz[i] * mz;
  out[7] = approx_atan_positive(half_h / k[4]) * 2.0f;  // fov_h from fy
  out[8] = approx_atan_positive(half_w / k[0]) * 2.0f;  // fov_w from fx
}

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RESID_LS = 2, ACT_RELU = 3 };

// y[v][n] = act(bias[n] + sum_k x[v][k] * w[n][k]) for v < V. ACT_RESID_LS: y[v][n] += gamma[n] * (...)   (y is the residual stream)
template <int ACT>
__device__ void wg_linear(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                          float* __restrict__ y, int ldy, int N, int K, int V, const float* __restrict__ gamma) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int n = wave; n < N; n += nwaves) {
    const float* wr = w + (long)n * K;
    float acc[kMaxViews];
#pragma unroll
    for (int v = 0; v < kMaxViews; ++v) acc[v] = 0.f;
    if ((K & 3) == 0) {
      for (int k = lane * 4; k < K; k += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
        for (int v = 0; v < kMaxViews; ++v)
          if (v < V) {
            const float4 xv = *reinterpret_cast<const float4*>(x + (long)v * ldx + k);
            acc[v] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
          }
      }
    } else {
      for (int k = lane; k < K; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int v = 0; v < kMaxViews; ++v)
          if (v < V) acc[v] += wv * x[(long)v * ldx + k];
      }
    }
#pragma unroll
    for (int v = 0; v < kMaxViews; ++v)
      if (v < V) {
        float t = wave_sum(acc[v]);
        if (lane == 0) {
          t += bias[n];
          if (ACT == ACT_GELU) t = gelu_erf(t);
          if (ACT == ACT_RELU) t = fmaxf(t, 0.f);
          if (ACT == ACT_RESID_LS) t = y[(long)v * ldy + n] + gamma[n] * t;
          y[(long)v * ldy + n] = t;
        }
      }
  }
}

// out[v][:] = LayerNorm(x[v][:]) (biased variance, like Burn's LayerNorm)
__device__ void wg_layernorm(const float* __restrict__ x, float* __restrict__ out, int V, int D, const float* __restrict__ g,
                             const float* __restrict__ b, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int v = wave; v < V; v += nwaves) {
    const float* r = x + (long)v * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += r[k];
    const float mean = wave_sum(s) / D;
    float q = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float c = r[k] - mean;
      q += c * c;
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    for (int k = lane; k < D; k += 64) out[(long)v * D + k] = (r[k] - mean) * rstd * g[k] + b[k];
  }
}

__global__ __launch_bounds__(1024) void camera_encoder_kernel(const float* __restrict__ extr, const float* __restrict__ intr, int V, int D,
                                                              int heads, float half_h, float half_w, float eps_tok, float eps_blk,
                                                              CamEncW w, float* __restrict__ scratch, float* __restrict__ out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int hd = D / heads;
  // per-image scratch: pose [V][12] | x [V][D] | xn [V][D] | qkv [V][3D] | hid [V][4D]
  float* pose = scratch + (long)b * V * (12 + 9 * (long)D);
  float* x = pose + V * 12;
  float* xn = x + (long)V * D;
  float* qkv = xn + (long)V * D;
  float* hid = qkv + (long)V * 3 * D;
  if (tid < V) pose_encode(extr + ((long)b * V + tid) * 12, intr + ((long)b * V + tid) * 9, half_h, half_w, pose + tid * 12);
  __syncthreads();
  wg_linear<ACT_GELU>(pose, 12, w.fc1_w, w.fc1_b, hid, 4 * D, D / 2, 9, V, nullptr);
  __syncthreads();
  wg_linear<ACT_NONE>(hid, 4 * D, w.fc2_w, w.fc2_b, xn, D, D, D / 2, V, nullptr);
  __syncthreads();
  wg_layernorm(xn, x, V, D, w.tn_g, w.tn_b, eps_tok);
  __syncthreads();
  for (int i = 0; i < w.depth; ++i) {
    const CamEncW::Blk& k = w.blk[i];
    wg_layernorm(x, xn, V, D, k.n1g, k.n1b, eps_blk);
    __syncthreads();
    wg_linear<ACT_NONE>(xn, D, k.qkv_w, k.qkv_b, qkv, 3 * D, 3 * D, D, V, nullptr);
    __syncthreads();
    // attention over the V view tokens: one thread per (head, query); rows are q | k | v, head h at columns h*hd
    const float scale = rsqrtf((float)hd);
    for (int t = tid; t < heads * V; t += blockDim.x) {
      const int h = t / V, qi = t - h * V;
      const float* q = qkv + (long)qi * 3 * D + h * hd;
      float sc[kMaxViews], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < kMaxViews; ++j)
        if (j < V) {
          const float* kk = qkv + (long)j * 3 * D + D + h * hd;
          float a = 0.f;
          for (int e = 0; e < hd; ++e) a += q[e] * scale * kk[e];
          sc[j] = a;
          mx = fmaxf(mx, a);
        }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxViews; ++j)
        if (j < V) {
          sc[j] = expf(sc[j] - mx);
          den += sc[j];
        }
      const float inv = 1.0f / den;
      for (int e = 0; e < hd; ++e) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxViews; ++j)
          if (j < V) a += sc[j] * qkv[(long)j * 3 * D + 2 * D + h * hd + e];
        xn[(long)qi * D + h * hd + e] = a * inv;
      }
    }
    __syncthreads();
    wg_linear<ACT_RESID_LS>(xn, D, k.proj_w, k.proj_b, x, D, D, D, V, k.ls1);
    __syncthreads();
    wg_layernorm(x, xn, V, D, k.n2g, k.n2b, eps_blk);
    __syncthreads();
    wg_linear<ACT_GELU>(xn, D, k.fc1_w, k.fc1_b, hid, 4 * D, 4 * D, D, V, nullptr);
    __syncthreads();
    wg_linear<ACT_RESID_LS>(hid, 4 * D, k.fc2_w, k.fc2_b, x, D, D, 4 * D, V, k.ls2);
    __syncthreads();
  }
  wg_layernorm(x, xn, V, D, w.on_g, w.on_b, eps_tok);
  __syncthreads();
  for (int c = tid; c < D; c += blockDim.x) {  // tokens.mean_dim(1)
    float a = 0.f;
    for (int v = 0; v < V; ++v) a += xn[(long)v * D + c];
    out[(long)b * D + c] = a / V;
  }
}

// ---- camera DECODER (`CameraDecoder::forward`, camera.rs:143-199): two din x din linears + ReLU on the [B, din] camera feature, then
//      three small heads and the pose -> matrices tail. A din x din fp32 weight is 2.4 MB (din = 768): one workgroup streams it in
//      ~20 us, 48 workgroups of 16 output columns each in ~2 -- so the two wide layers are one multi-workgroup launch each (every
//      wave owns output columns, lanes stride K, the B rows ride in registers) and the heads + tail are one single-workgroup launch.
template <int ACT>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int ldy, int N, int K, int B,
                                                          int cols_per_wg) {
  const int n0 = blockIdx.x * cols_per_wg;
  const int n1 = n0 + cols_per_wg < N ? n0 + cols_per_wg : N;
  for (int b0 = 0; b0 < B; b0 += kMaxViews) {
    const int V = B - b0 < kMaxViews ? B - b0 : kMaxViews;
    wg_linear<ACT>(x + (long)b0 * ldx, ldx, w + (long)n0 * K, bias + n0, y + (long)b0 * ldy + n0, ldy, n1 - n0, K, V, nullptr);
  }
}

__global__ __launch_bounds__(256) void camera_heads_kernel(const float* __restrict__ h, int din, const float* __restrict__ wt,
                                                           const float* __restrict__ bt, const float* __restrict__ wq, const float* __restrict__ bq,
                                                           const float* __restrict__ wf, const float* __restrict__ bf, int H, int W,
                                                           float* __restrict__ pose, float* __restrict__ extr, float* __restrict__ intr) {
  // pose = (t3 | quat4 | relu(fov2)) of image blockIdx.x (camera.rs:171-181), then pose_encoding_to_extri_intri (camera.rs:281-358)
  const int b = blockIdx.x;
  const float* hb = h + (long)b * din;
  float* pb = pose + b * 9;
  wg_linear<ACT_NONE>(hb, din, wt, bt, pb, 9, 3, din, 1, nullptr);
  wg_linear<ACT_NONE>(hb, din, wq, bq, pb + 3, 9, 4, din, 1, nullptr);
  wg_linear<ACT_RELU>(hb, din, wf, bf, pb + 7, 9, 2, din, 1, nullptr);
  __syncthreads();
  if (threadIdx.x == 0 && (extr || intr)) pose_to_camera_one(pb, H, W, extr ? extr + b * 12 : nullptr, intr ? intr + b * 9 : nullptr);
}

}  // namespace

int launch_camera_decoder(const float* cam, int B, int din, const CamDecW& w, int H, int W, float* h1, float* h2, float* pose, float* extr,
                          float* intr, hipStream_t s) {
  if (din % 4 != 0 || B < 1) MD_FAIL(MD_ERR_INVALID_ARG, "camera decoder: width %d, batch %d", din, B);
  const int cols = 16, grid = (din + cols - 1) / cols;
  hipLaunchKernelGGL(linear_rows_kernel<ACT_RELU>, dim3(grid), dim3(256), 0, s, cam, din, w.w1, w.b1, h1, din, din, din, B, cols);
  hipLaunchKernelGGL(linear_rows_kernel<ACT_RELU>, dim3(grid), dim3(256), 0, s, (const float*)h1, din, w.w2, w.b2, h2, din, din, din, B, cols);
  hipLaunchKernelGGL(camera_heads_kernel, dim3(B), dim3(256), 0, s, (const float*)h2, din, w.wt, w.bt, w.wq, w.bq, w.wf, w.bf, H, W, pose, extr, intr);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

size_t camera_encoder_scratch_floats(int B, int V, int D) { return (size_t)B * V * (12 + 9 * (size_t)D); }

int launch_camera_encoder(const float* extr, const float* intr, int B, int V, int D, int heads, int H, int W, float eps_tok, float eps_blk,
                          const CamEncW& w, float* scratch, float* out, hipStream_t s) {
  if (V < 1 || V > kMaxViews) MD_FAIL(MD_ERR_UNSUPPORTED, "camera encoder: %d views (1..%d supported)", V, kMaxViews);
  if (heads <= 0 || D % heads != 0 || D % 8 != 0 || w.depth < 0 || w.depth > CamEncW::kMaxDepth)
    MD_FAIL(MD_ERR_INVALID_ARG, "camera encoder: width %d, %d heads, depth %d", D, heads, w.depth);
  hipLaunchKernelGGL(camera_encoder_kernel, dim3(B), dim3(1024), 0, s, extr, intr, V, D, heads, H * 0.5f, W * 0.5f, eps_tok, eps_blk, w,
                     scratch, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

}  // namespace md
