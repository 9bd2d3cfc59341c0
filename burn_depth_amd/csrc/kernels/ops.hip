// HBM-bound and small kernels of the Depth Pro path for gfx950: every thread moves 16 bytes
// where the layout allows it, blocks are 256 threads (4 waves of 64), grids are capped and
// grid-strided (MI355X: 256 CUs x 8 blocks).
#include "ops.h"
#include "camera_math.h"
#include "elem.h"

namespace md {

static inline int grid_for(long work_items, int block = 256, int cap = 256 * 8) {
  long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// a1 rgb_to_input_tensor (src/inference.rs:103-111)
// ------------------------------------------------------------------------------------------------
__global__ void rgb_to_input_kernel(const uint8_t* __restrict__ rgb, long hw, float* __restrict__ out) {
#pragma clang fp contract(off)
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (float)rgb[i * 3 + c] / 255.0f;
      out[c * hw + i] = (v - mean[c]) / sd[c];
    }
  }
}

int launch_rgb_to_input(const uint8_t* rgb, int w, int h, float* out, hipStream_t s) {
  long hw = (long)w * h;
  hipLaunchKernelGGL(rgb_to_input_kernel, dim3(grid_for(hw)), dim3(256), 0, s, rgb, hw, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// a2 bilinear sampling (interpolate.rs:29-52, 78-89). fp contraction off: bit-exact with the
// reference's separate f32 multiplies and adds.
// ------------------------------------------------------------------------------------------------
struct AxisTap {
  int i0, i1;
  float d;
};

__device__ __forceinline__ AxisTap axis_tap(int o, int in_size, int out_size, int method) {
#pragma clang fp contract(off)
  AxisTap t;
  if (method == MD_INTERP_CUSTOM) {
    const float scale = (float)in_size / (float)out_size;
    const float src = ((float)o + 0.5f) * scale - 0.5f;
    const float f0 = floorf(src);
    const float f1 = fminf(f0 + 1.0f, (float)(in_size - 1));
    t.i0 = (int)fmaxf(f0, 0.0f);
    t.i1 = (int)f1;
    t.d = src - f0;
  } else {  // align_corners = true (Burn module::interpolate)
    float src = 0.f;
    if (out_size > 1) src = (float)o * ((float)(in_size - 1) / (float)(out_size - 1));
    const float f0 = floorf(src);
    int i0 = (int)f0;
    i0 = i0 < 0 ? 0 : (i0 > in_size - 1 ? in_size - 1 : i0);
    t.i0 = i0;
    t.i1 = i0 + 1 > in_size - 1 ? in_size - 1 : i0 + 1;
    t.d = src - f0;
  }
  return t;
}

// axis_tap with the axis scale (in / out for the half-pixel form, (in-1) / (out-1) for align_corners) computed once on the
// host: the same correctly rounded fp32 quotient, without the ~10-instruction division sequence per thread and axis.
__host__ __device__ inline float axis_scale(int in_size, int out_size, int method) {
  if (method == MD_INTERP_CUSTOM) return (float)in_size / (float)out_size;
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}
__device__ __forceinline__ AxisTap axis_tap_s(int o, int in_size, float scale, int method) {
#pragma clang fp contract(off)
  AxisTap t;
  if (method == MD_INTERP_CUSTOM) {
    const float src = ((float)o + 0.5f) * scale - 0.5f;
    const float f0 = floorf(src);
    const float f1 = fminf(f0 + 1.0f, (float)(in_size - 1));
    t.i0 = (int)fmaxf(f0, 0.0f);
    t.i1 = (int)f1;
    t.d = src - f0;
  } else {
    const float src = (float)o * scale;  // out_size == 1: scale = 0
    const float f0 = floorf(src);
    int i0 = (int)f0;
    i0 = i0 < 0 ? 0 : (i0 > in_size - 1 ? in_size - 1 : i0);
    t.i0 = i0;
    t.i1 = i0 + 1 > in_size - 1 ? in_size - 1 : i0 + 1;
    t.d = src - f0;
  }
  return t;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ plane, int iw, const AxisTap& ty, const AxisTap& tx) {
#pragma clang fp contract(off)
  const float tl = plane[(long)ty.i0 * iw + tx.i0], tr = plane[(long)ty.i0 * iw + tx.i1];
  const float bl = plane[(long)ty.i1 * iw + tx.i0], br = plane[(long)ty.i1 * iw + tx.i1];
  const float top = tl * (1.0f - tx.d) + tr * tx.d;
  const float bottom = bl * (1.0f - tx.d) + br * tx.d;
  return top * (1.0f - ty.d) + bottom * ty.d;
}

// One workgroup per output row segment: blockIdx.x = plane * OH + oy (one 32-bit division per workgroup), blockIdx.y = a
// 1024-column chunk; a thread owns columns tid, tid + 256, ... of the chunk, so every load and store instruction of a wave
// walks consecutive columns (a thread owning 4 ADJACENT columns measured 0.73x on the 2:1 downscale: its loads then
// stride 32 bytes across the wave). The first form (grid-stride over output elements) spent its time in three 64-bit
// integer divisions per element: [8,1,1536^2] -> 1080x1920 1.5 -> 3.5 TB/s.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int planes, int H, int W,
                                                              float* __restrict__ out, int OH, int OW, int method, int post,
                                                              float sy, float sx) {
  const int row = blockIdx.x;
  const int pl = row / OH, oy = row - pl * OH;
  const AxisTap ty = axis_tap_s(oy, H, sy, method);
  const float* plane = in + (long)pl * H * W;
  float* orow = out + (long)row * OW;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = blockIdx.y * 1024 + j * 256 + threadIdx.x;
    if (ox < OW) {
      const AxisTap tx = axis_tap_s(ox, W, sx, method);
      float r = bilerp(plane, W, ty, tx);
      if (post == 1) r = 1.0f / fminf(fmaxf(r, 1e-4f), 1e4f);
      orow[ox] = r;
    }
  }
}

__global__ void copy_post_kernel(const float* __restrict__ in, long n, float* __restrict__ out, int post) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = in[i];
    if (post == 1) v = 1.0f / fminf(fmaxf(v, 1e-4f), 1e4f);
    out[i] = v;
  }
}

int launch_resize_bilinear(const float* in, int planes, int H, int W, float* out, int OH, int OW, int method,
                           int post, hipStream_t s) {
  if (OH <= 0 || OW <= 0) MD_FAIL(MD_ERR_SHAPE, "output size must be positive");
  if (H == OH && W == OW) {  // identity (interpolate.rs:61-63)
    const long n = (long)planes * H * W;
    if (in != out || post)
      hipLaunchKernelGGL(copy_post_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, n, out, post);
  } else {
    const long rows = (long)planes * OH;
    if (rows > 0x7fffffffL || OW > 65535 * 1024) MD_FAIL(MD_ERR_UNSUPPORTED, "resize: %ld output rows of %d columns", rows, OW);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)rows, (unsigned)((OW + 1023) / 1024)), dim3(256), 0, s, in, planes,
                       H, W, out, OH, OW, method, post, axis_scale(H, OH, method), axis_scale(W, OW, method));
  }
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// a2+a3 fused pyramid + split + patchify -> A matrix of the patch-embed GEMM
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pyramid_patchify_kernel(const float* __restrict__ x, PyramidGeom g, T* __restrict__ out) {
  const int grid = g.win / g.ps;             // patches per tile side
  const int P = grid * grid;
  const int K = 3 * g.ps * g.ps;
  const int K8 = K / 8;
  const int n0 = g.steps0 * g.steps0 * g.B, n1 = g.steps1 * g.steps1 * g.B;
  const long total = (long)(n0 + n1 + g.B) * P * K8;
  const int S = g.S;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int col = (int)(e % K8) * 8;
    const long row = e / K8;
    const int pidx = (int)(row % P);
    const int tile = (int)(row / P);
    const int c = col / (g.ps * g.ps);
    const int ky = (col / g.ps) % g.ps;
    const int kx = col % g.ps;
    const int py = pidx / grid, px = pidx % grid;
    const int ty = py * g.ps + ky, tx = px * g.ps + kx;
    int level, b, j, i, stride;
    if (tile < n0) {
      level = 0; b = tile % g.B; int ji = tile / g.B; j = ji / g.steps0; i = ji % g.steps0; stride = g.stride0;
    } else if (tile < n0 + n1) {
      level = 1; int t = tile - n0; b = t % g.B; int ji = t / g.B; j = ji / g.steps1; i = ji % g.steps1; stride = g.stride1;
    } else {
      level = 2; b = tile - n0 - n1; j = 0; i = 0; stride = 0;
    }
    const int Y = j * stride + ty, X = i * stride + tx;
    const float* plane = x + ((long)b * 3 + c) * (long)S * S;
    float v[8];
    if (level == 0) {
      const f32x4_t a0 = *(const f32x4_t*)(plane + (long)Y * S + X);
      const f32x4_t a1 = *(const f32x4_t*)(plane + (long)Y * S + X + 4);
      v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3];
      v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
    } else {
      const int SL = level == 1 ? S / 2 : S / 4;  // compute_output_size(S, .5|.25), S % 4 == 0
      const AxisTap tyy = axis_tap(Y, S, SL, g.method);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const AxisTap txx = axis_tap(X + q, S, SL, g.method);
        v[q] = bilerp(plane, S, tyy, txx);
      }
    }
    if constexpr (is_split<T>::value) {  // split-half rows [hi: K | lo: K]
      i32x4_t hi, lo;
      split8<T>((f32x4_t){v[0], v[1], v[2], v[3]}, (f32x4_t){v[4], v[5], v[6], v[7]}, hi, lo);
      *(i32x4_t*)(out + row * 2 * K + col) = hi;
      *(i32x4_t*)(out + row * 2 * K + K + col) = lo;
    } else {
      store8<T>(out + row * K + col, v);
    }
  }
}

// The same result from ONE read of the image (InterpolationMethod::Custom, patch size 16): a workgroup owns a
// 64 x 64 block of one channel plane (= 4 x 4 level-0 patches, 2 x 2 level-1 patches, 1 level-2 patch; every tile origin
// of every level is a multiple of the patch size, so patches never straddle blocks). The block is staged once in LDS
// with whole 256-byte row segments; each (patch, channel) is then ONE contiguous run of 256 elements of an A-matrix row
// and a wave writes it with one instruction per destination tile (overlapping windows: up to 2 x 2 tiles hold a patch).
// The x0.5 / x0.25 taps of align_corners=False land inside the block (src = 2o + 0.5 and 4o + 1.5), and the values go
// through the same axis_tap / bilerp arithmetic as the stand-alone resize: bit-identical to the generic kernel.
template <typename T>
__global__ __launch_bounds__(256) void pyramid_patchify_blocks_kernel(const float* __restrict__ x, PyramidGeom g, T* __restrict__ out) {
  constexpr int BS = 64, LD = 68;  // LDS row stride 68 floats: the 16-byte row reads of a 16-row patch spread over the banks
  __shared__ __attribute__((aligned(16))) float blk[BS * LD];
  const int S = g.S, nb = S / BS;           // blocks per image side
  const int grid = g.win / 16, P = grid * grid, K = 3 * 256;
  const int st0 = g.stride0 / 16, st1 = g.stride1 / 16;  // tile strides in patches
  int id = blockIdx.x;
  const int bx = id % nb; id /= nb;
  const int by = id % nb; id /= nb;
  const int c = id % 3;
  const int b = id / 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* plane = x + ((long)b * 3 + c) * (long)S * S + (long)by * BS * S + bx * BS;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // 16 lanes per 256-byte row segment, 4 rows per wave-instruction
    const int row = (i * 4 + wave) * 4 + (lane >> 4), col = (lane & 15) * 4;
    *(f32x4_t*)(blk + row * LD + col) = *(const f32x4_t*)(plane + (long)row * S + col);
  }
  __syncthreads();
  const int n0 = g.steps0 * g.steps0 * g.B, n1 = g.steps1 * g.steps1 * g.B;
  const int ky = lane >> 2, kx = (lane & 3) * 4;  // the lane's 4 pixels of a 16 x 16 patch
  // destination rows of global patch (gy, gx) of a level: every tile (j, i) of that level whose window holds it
  auto emit = [&](int level, int gy, int gx, const float* v) __attribute__((always_inline)) {
    const int steps = level == 0 ? g.steps0 : (level == 1 ? g.steps1 : 1);
    const int stp = level == 0 ? st0 : (level == 1 ? st1 : 0);
    const int base = level == 0 ? 0 : (level == 1 ? n0 : n0 + n1);
    int j0 = 0, j1 = 0, i0 = 0, i1 = 0;
    if (steps > 1) {
      j1 = gy / stp; j1 = j1 < steps - 1 ? j1 : steps - 1;            // last tile starting at or before gy
      j0 = gy - grid + stp >= 0 ? (gy - grid + stp) / stp : 0;        // first tile whose window [stp*j, stp*j + grid) holds gy
      i1 = gx / stp; i1 = i1 < steps - 1 ? i1 : steps - 1;
      i0 = gx - grid + stp >= 0 ? (gx - grid + stp) / stp : 0;
    }
    for (int j = j0; j <= j1; ++j)
      for (int i = i0; i <= i1; ++i) {
        const long tile = base + (long)(j * steps + i) * g.B + b;
        const long row = tile * P + (gy - j * stp) * grid + (gx - i * stp);
        store4p<T>(out + row * (K * kPlanes<T>) + c * 256 + ky * 16 + kx, K, (f32x4_t){v[0], v[1], v[2], v[3]});
      }
  };
  // level 0: 16 patches, 4 per wave (copies)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pi = wave * 4 + q, py = pi >> 2, px = pi & 3;
    const f32x4_t a = *(const f32x4_t*)(blk + (py * 16 + ky) * LD + px * 16 + kx);
    const float v[4] = {a[0], a[1], a[2], a[3]};
    emit(0, by * 4 + py, bx * 4 + px, v);
  }
  // level 1 (x0.5): 4 patches, one per wave; level 2 (x0.25): 1 patch, wave 0 again
  {
    const int py = wave >> 1, px = wave & 1;
    const int oy = by * 32 + py * 16 + ky;
    const AxisTap ty = axis_tap(oy, S, S / 2, MD_INTERP_CUSTOM);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const AxisTap tx = axis_tap(bx * 32 + px * 16 + kx + e, S, S / 2, MD_INTERP_CUSTOM);
      AxisTap ly = ty, lx = tx;  // block-local taps
      ly.i0 -= by * BS; ly.i1 -= by * BS; lx.i0 -= bx * BS; lx.i1 -= bx * BS;
      v[e] = bilerp(blk, LD, ly, lx);
    }
    emit(1, by * 2 + py, bx * 2 + px, v);
  }
  if (wave == 0) {
    const AxisTap ty = axis_tap(by * 16 + ky, S, S / 4, MD_INTERP_CUSTOM);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const AxisTap tx = axis_tap(bx * 16 + kx + e, S, S / 4, MD_INTERP_CUSTOM);
      AxisTap ly = ty, lx = tx;
      ly.i0 -= by * BS; ly.i1 -= by * BS; lx.i0 -= bx * BS; lx.i1 -= bx * BS;
      v[e] = bilerp(blk, LD, ly, lx);
    }
    emit(2, by, bx, v);
  }
}

int launch_pyramid_patchify(const float* x, const PyramidGeom& g, void* patches, int prec, hipStream_t s, bool force_generic) {
  if (g.ps % 8 != 0 || g.win % g.ps != 0 || g.S % 4 != 0)
    MD_FAIL(MD_ERR_UNSUPPORTED, "pyramid: patch %d / window %d / size %d unsupported", g.ps, g.win, g.S);
  // one-read block kernel: patch 16, whole 64-pixel blocks, tile strides on the patch grid, align_corners=False taps
  const bool blocks_ok = !force_generic && g.method == MD_INTERP_CUSTOM && g.ps == 16 && g.S % 64 == 0 && g.S == 4 * g.win && g.stride0 % 16 == 0 &&
                         g.stride1 % 16 == 0 && (g.steps0 - 1) * g.stride0 + g.win == g.S && (g.steps1 - 1) * g.stride1 + g.win == g.S / 2;
  if (blocks_ok) {
    const long nblocks = (long)g.B * 3 * (g.S / 64) * (g.S / 64);
    MD_BY_PREC(prec, hipLaunchKernelGGL(pyramid_patchify_blocks_kernel<T>, dim3((unsigned)nblocks), dim3(256), 0, s, x, g, (T*)patches));
    MD_HIP(hipGetLastError());
    return MD_OK;
  }
  const int grid = g.win / g.ps;
  const long total = (long)(g.steps0 * g.steps0 * g.B + g.steps1 * g.steps1 * g.B + g.B) * grid * grid *
                     (3 * g.ps * g.ps / 8);
  MD_BY_PREC(prec, hipLaunchKernelGGL(pyramid_patchify_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, g, (T*)patches));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// generic patch extraction + NHWC bilinear resize (Depth-Anything-v3 path)
// ------------------------------------------------------------------------------------------------
// `ci.x != nullptr`: the same launch also writes the cls rows (cls + pos[0]) and the zero padding rows of the fp32 residual
// stream (what cls_init_kernel does; DA3 runs both at the start of every frame and each launch costs its ~4.5 us floor)
struct PatchifyCls {
  float* x;
  int S, n_tokens, D;
  const float *cls, *pos0;
};
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ x, int B, int H, int W, int ps, int Kp, T* __restrict__ out, PatchifyCls ci) {
  const int ph = H / ps, pw = W / ps, K = 3 * ps * ps;
  const long total = (long)B * ph * pw * Kp;
  if (ci.x) {
    const int extra = 1 + (ci.S - ci.n_tokens);  // cls row + padding rows of every sequence
    const long ctotal = (long)B * extra * ci.D;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < ctotal; e += (long)gridDim.x * blockDim.x) {
      const int d = (int)(e % ci.D);
      const long t = e / ci.D;
      const int r = (int)(t % extra);
      const int seq = (int)(t / extra);
      if (r == 0) ci.x[((long)seq * ci.S) * ci.D + d] = ci.cls[d] + ci.pos0[d];
      else ci.x[((long)seq * ci.S + ci.n_tokens + r - 1) * ci.D + d] = 0.f;
    }
  }
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int col = (int)(e % Kp);
    const long row = e / Kp;
    float v = 0.f;
    if (col < K) {
      const int c = col / (ps * ps), r = col % (ps * ps);
      const int ky = r / ps, kx = r % ps;
      const int px = (int)(row % pw);
      const long t = row / pw;
      const int py = (int)(t % ph);
      const int b = (int)(t / ph);
      v = x[(((long)b * 3 + c) * H + py * ps + ky) * W + px * ps + kx];
    }
    st1p<T>(out + row * ((long)Kp * kPlanes<T>) + col, Kp, v);  // split-half rows: [hi: Kp | lo: Kp]
  }
}

int launch_patchify(const float* x, int B, int H, int W, int ps, int Kp, void* out, int prec, hipStream_t s, float* cls_x, int S,
                    int n_tokens, int D, const float* cls, const float* pos0) {
  if (ps <= 0 || H % ps || W % ps || Kp < 3 * ps * ps) MD_FAIL(MD_ERR_SHAPE, "patchify: %dx%d / patch %d / K %d", H, W, ps, Kp);
  if (cls_x && (!cls || !pos0 || S < n_tokens || D <= 0)) MD_FAIL(MD_ERR_INVALID_ARG, "patchify: cls rows need the cls token and pos[0]");
  const long total = (long)B * (H / ps) * (W / ps) * Kp;
  const PatchifyCls ci{cls_x, S, n_tokens, D, cls, pos0};
  MD_BY_PREC(prec, hipLaunchKernelGGL(patchify_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, B, H, W, ps, Kp, (T*)out, ci));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// One workgroup per output row (blockIdx.x = b * OH + oy: one 32-bit division per workgroup); a thread = 8 channels of one
// output pixel (16-byte bf16 / 32-byte f32 accesses), threads walk (column, channel group) with the channel group fastest.
// The first form was a grid-stride loop over elements with four 64-bit integer divisions per thread and iteration.
template <typename T>
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const T* __restrict__ in, int B, int H, int W, int C, long ld_in,
                                                          T* __restrict__ out, int OH, int OW, long ld_out, int method,
                                                          const float* __restrict__ addend, int c8_shift, float sy, float sx) {
  const int C8 = C >> 3;
  const int row = blockIdx.x;
  const int b = row / OH, oy = row - b * OH;
  const AxisTap ty = axis_tap_s(oy, H, sy, method);
  constexpr int PL = kPlanes<T>;  // split-half pixels: [hi: ld | lo: ld]
  const long pin = ld_in * PL, pout = ld_out * PL;
  const T* r0 = in + ((long)b * H + ty.i0) * W * pin;
  const T* r1 = in + ((long)b * H + ty.i1) * W * pin;
  T* orow = out + (long)row * OW * pout;
  const float* arow = addend ? addend + (long)oy * OW * C : nullptr;
  const int n = OW * C8;
  for (int e = blockIdx.y * 256 + threadIdx.x; e < n; e += gridDim.y * 256) {
    const int ox = c8_shift >= 0 ? e >> c8_shift : e / C8;
    const int c = (e - ox * C8) * 8;
    const AxisTap tx = axis_tap_s(ox, W, sx, method);
    float tl[8], tr[8], bl[8], br[8], o[8];
    load8fp<T>(r0 + (long)tx.i0 * pin + c, ld_in, tl);
    load8fp<T>(r0 + (long)tx.i1 * pin + c, ld_in, tr);
    load8fp<T>(r1 + (long)tx.i0 * pin + c, ld_in, bl);
    load8fp<T>(r1 + (long)tx.i1 * pin + c, ld_in, br);
    if constexpr (sizeof(T) == 4 || is_split<T>::value) {  // fp32 parity mode and the split-half accurate mode: the reference's separate multiplies and adds (interpolate.rs:78-89)
#pragma clang fp contract(off)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float top = tl[i] * (1.0f - tx.d) + tr[i] * tx.d;
        const float bottom = bl[i] * (1.0f - tx.d) + br[i] * tx.d;
        o[i] = top * (1.0f - ty.d) + bottom * ty.d;
      }
    } else {  // 16-bit storage: three fused lerps (6 instead of 9 operations per channel); the difference to the form above
              // is an fp32 ulp, far below the output rounding -- this kernel is VALU-bound, not HBM-bound
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float top = fmaf(tx.d, tr[i] - tl[i], tl[i]);
        const float bottom = fmaf(tx.d, br[i] - bl[i], bl[i]);
        o[i] = fmaf(ty.d, bottom - top, top);
      }
    }
    if (arow) {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += arow[(long)ox * C + c + i];
    }
    store8p<T>(orow + (long)ox * pout + c, ld_out, o);
  }
}

template <typename T>
__global__ void border_bias_fix_kernel(T* __restrict__ map, int B, int H, int W, int C, long ld, const float* __restrict__ bias9) {
  const int nb = 2 * W + 2 * H - 4;  // border pixels of one image: top row, bottom row, then the two columns without corners
  const long total = (long)B * nb * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int t = (int)(e / C);
    const int k = t % nb, b = t / nb;
    int Y, X;
    if (k < W) { Y = 0; X = k; }
    else if (k < 2 * W) { Y = H - 1; X = k - W; }
    else if (k < 2 * W + H - 2) { Y = k - 2 * W + 1; X = 0; }
    else { Y = k - (2 * W + H - 2) + 1; X = W - 1; }
    const int cls = (Y == 0 ? 0 : (Y == H - 1 ? 2 : 1)) * 3 + (X == 0 ? 0 : (X == W - 1 ? 2 : 1));
    T* p = map + (((long)b * H + Y) * W + X) * (ld * kPlanes<T>) + c;  // split-half rows: [hi: ld | lo: ld]
    st1p<T>(p, ld, ld1p<T>(p, ld) + (bias9[cls * C + c] - bias9[4 * C + c]));
  }
}

int launch_border_bias_fix(void* map, int B, int H, int W, int C, long ld, const float* bias9, int prec, hipStream_t s) {
  if (H < 2 || W < 2) MD_FAIL(MD_ERR_UNSUPPORTED, "border bias fix: %dx%d map", H, W);
  const long total = (long)B * (2 * W + 2 * H - 4) * C;
  MD_BY_PREC(prec, hipLaunchKernelGGL(border_bias_fix_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (T*)map, B, H, W, C, ld, bias9));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int launch_resize_nhwc(const void* in, int B, int H, int W, int C, long ld_in, void* out, int OH, int OW, long ld_out,
                       int method, const float* addend, int prec, hipStream_t s) {
  if (C % 8 != 0 || OH <= 0 || OW <= 0) MD_FAIL(MD_ERR_UNSUPPORTED, "resize_nhwc: C=%d must be a multiple of 8", C);
  const long rows = (long)B * OH;
  if (rows > 0x7fffffffL) MD_FAIL(MD_ERR_UNSUPPORTED, "resize_nhwc: %ld output rows", rows);
  const int c8 = C / 8;
  int c8_shift = -1;
  for (int k = 0; k < 16; ++k)
    if ((1 << k) == c8) c8_shift = k;
  const int per_row = OW * c8;
  const int gy = std::max(1, std::min(8, (per_row + 4 * 256 - 1) / (4 * 256)));  // ~4 pixels-groups per thread
  MD_BY_PREC(prec, hipLaunchKernelGGL(resize_nhwc_kernel<T>, dim3((unsigned)rows, (unsigned)gy), dim3(256), 0, s, (const T*)in, B, H, W, C, ld_in, (T*)out, OH, OW, ld_out, method, addend, c8_shift, axis_scale(H, OH, method), axis_scale(W, OW, method)));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// a3 split / a6 merge on fp32 NCHW (stand-alone ops and debug taps)
// ------------------------------------------------------------------------------------------------
__global__ void split_kernel(const float* __restrict__ x, int B, int C, int S, int win, int stride, int steps,
                             float* __restrict__ out) {
  const long total = (long)steps * steps * B * C * win * win;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tx = (int)(e % win);
    long t = e / win;
    const int ty = (int)(t % win);
    t /= win;
    const int c = (int)(t % C);
    t /= C;
    const int b = (int)(t % B);
    const int ji = (int)(t / B);
    const int j = ji / steps, i = ji % steps;
    out[e] = x[(((long)b * C + c) * S + j * stride + ty) * S + i * stride + tx];
  }
}

int launch_split(const float* x, int B, int C, int S, int win, int stride, int steps, float* out, hipStream_t s) {
  const long total = (long)steps * steps * B * C * win * win;
  hipLaunchKernelGGL(split_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, B, C, S, win, stride, steps, out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

__global__ void merge_kernel(const float* __restrict__ tiles, int B, int C, int h, int w, int steps, int pad,
                             float* __restrict__ out, int OH, int OW) {
  const long total = (long)B * C * OH * OW;
  const int ih = h - 2 * pad, iw = w - 2 * pad;  // interior extent
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int X = (int)(e % OW);
    long t = e / OW;
    const int Y = (int)(t % OH);
    t /= OH;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    int j = 0, i = 0;
    if (steps > 1) {
      j = Y < pad ? 0 : (Y - pad) / ih;
      j = j > steps - 1 ? steps - 1 : j;
      i = X < pad ? 0 : (X - pad) / iw;
      i = i > steps - 1 ? steps - 1 : i;
    }
    const int ty = Y - j * ih, tx = X - i * iw;
    const long tile = (long)(j * steps + i) * B + b;
    out[e] = tiles[((tile * C + c) * h + ty) * w + tx];
  }
}

int launch_merge(const float* tiles, int B, int C, int h, int w, int steps, int pad, float* out, int OH, int OW,
                 hipStream_t s) {
  const long total = (long)B * C * OH * OW;
  hipLaunchKernelGGL(merge_kernel, dim3(grid_for(total)), dim3(256), 0, s, tiles, B, C, h, w, steps, pad, out, OH,
                     OW);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// cls token row + zero padding rows of the residual stream
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int group_of_seq(const SeqGroups& g, int seq) {
  int r = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < g.ngroups && seq >= g.seq0[i]) r = i;
  return r;
}

#define MD_SEL4(arr, g) ((g) == 0 ? (arr)[0] : (g) == 1 ? (arr)[1] : (g) == 2 ? (arr)[2] : (arr)[3])

__global__ void cls_init_kernel(float* __restrict__ x, int nseq, int S, int n_tokens, int D, SeqGroups g) {
  const int extra = 1 + (S - n_tokens);  // cls row + padding rows
  const long total = (long)nseq * extra * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    long t = e / D;
    const int r = (int)(t % extra);
    const int seq = (int)(t / extra);
    const int gi = group_of_seq(g, seq);
    if (r == 0) {
      const float* cls = MD_SEL4(g.a, gi);
      const float* pos = MD_SEL4(g.b, gi);
      x[((long)seq * S) * D + d] = cls[d] + pos[d];
    } else {
      x[((long)seq * S + n_tokens + r - 1) * D + d] = 0.f;
    }
  }
}

int launch_cls_init(float* x, int nseq_total, int S, int n_tokens, int D, const SeqGroups& g, hipStream_t s) {
  const long total = (long)nseq_total * (1 + S - n_tokens) * D;
  hipLaunchKernelGGL(cls_init_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, nseq_total, S, n_tokens, D, g);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// K3 LayerNorm: one wave per row, row held in registers (D <= 1024), two-pass statistics in fp32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// four values -> four saturating OCP e4m3 bytes
__device__ __forceinline__ int pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.f, 448.f), __builtin_amdgcn_fmed3f(b, -448.f, 448.f), 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.f, 448.f), __builtin_amdgcn_fmed3f(d, -448.f, 448.f), w, true);
}

// `tok0` != nullptr: row 0 of every sequence is REPLACED first by tok0[seq * tok0_stride ..] (DA3: the camera token takes the cls
// slot at the first extended block; this kernel is that block's first LayerNorm) -- the new row is normalised and written back to x
template <typename TO, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, TO* __restrict__ out, long rows,
                                                        int D, float eps, int S, SeqGroups g, float fp8_inv_scale,
                                                        const float* __restrict__ tok0, int tok0_stride, float* __restrict__ x_rw) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  // the row a wave works on was requested one iteration earlier, BEFORE the previous row's stores: vmcnt retires loads and stores in order, so
  // a row's loads issued behind the previous row's stores would wait for those stores' acknowledgements (MD_LN_NO_PREFETCH: the plain loop)
  auto row_src = [&](long row, unsigned& seq, bool& repl) __attribute__((always_inline)) {
    // 32-bit division (the launcher holds rows below 2^31): the 64-bit form is ~170 vector instructions per row
    seq = (unsigned)row / (unsigned)S;
    repl = tok0 != nullptr && (unsigned)row - seq * (unsigned)S == 0u;  // wave-uniform
    return repl ? tok0 + (long)seq * tok0_stride : x + row * D;
  };
  auto load_row = [&](const float* xr, f32x4_t (&v)[NV]) __attribute__((always_inline)) {
    // every load of the row in one straight-line block: the token-0 write-back used to sit INSIDE this loop (round 4), which cut it into four
    // conditional blocks -- one load in flight per wave instead of four -- and cost the Depth Pro step 4 ms (LayerNorm 197 -> 277 us per
    // launch, 5.3 -> 3.8 TB/s: what the round-4 review saw as a 9.9 / 13.7 ms "discrepancy" between two of that round's bench records)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane * 4 + k * 256;
      v[k] = i < D ? *(const f32x4_t*)(xr + i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4_t v[NV], vn[NV];
  unsigned seq = 0, seq_n = 0;
  bool repl = false, repl_n = false;
  if (wave_id < rows) load_row(row_src(wave_id, seq, repl), v);
  for (long row = wave_id; row < rows; row += nwaves) {
#ifndef MD_LN_NO_PREFETCH
    const long next = row + nwaves;
    if (next < rows) load_row(row_src(next, seq_n, repl_n), vn);
#else
    if (row != wave_id) load_row(row_src(row, seq, repl), v);
#endif
    float sum = 0.f;
    if (repl) {  // wave-uniform, one row per sequence of ONE launch per frame
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane * 4 + k * 256;
        if (i < D) *(f32x4_t*)(x_rw + row * D + i) = v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane * 4 + k * 256;
      if (i < D) {
        const f32x4_t c = v[k] - mean;
        sq += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
    const int gi = group_of_seq(g, (int)seq);
    const float* gamma = MD_SEL4(g.a, gi);
    const float* beta = MD_SEL4(g.b, gi);
    // gamma / beta of the whole row before the first store: a load issued behind a store waits for that store's acknowledgement (vmcnt counts
    // loads and stores together, in order), which put one store round trip between the row's four output vectors
    f32x4_t gv[NV], bv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane * 4 + k * 256;
      const bool in = gamma != nullptr && i < D;
      gv[k] = in ? *(const f32x4_t*)(gamma + i) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
      bv[k] = in ? *(const f32x4_t*)(beta + i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // every output vector of the row is COMPUTED before the first store is issued, into registers of its own: in the per-vector form hipcc's
    // code had an `s_waitcnt vmcnt(0)` in front of each of the row's stores (one register pair held the packed data of all four; vmcnt
    // retires loads and stores in order, so each wait was the previous store's acknowledgement) -- three store round trips per row
    f32x4_t y[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      y[k] = (v[k] - mean) * rstd;
      if (gamma) y[k] = y[k] * gv[k] + bv[k];
    }
    if constexpr (sizeof(TO) == 4) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane * 4 + k * 256;
        if (i < D) *(f32x4_t*)((float*)out + row * D + i) = y[k];
      }
    } else if constexpr (sizeof(TO) == 1) {  // e4m3 MFMA operand on a static scale
      int w[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const f32x4_t q = y[k] * fp8_inv_scale;
        w[k] = pack4_fp8(q[0], q[1], q[2], q[3]);
        asm volatile("" : "+v"(w[k]));
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane * 4 + k * 256;
        if (i < D) *(int*)((char*)out + row * D + i) = w[k];
      }
    } else {
      i32x2_t hi[NV], lo[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if constexpr (is_split<TO>::value) {
          split4<TO>(y[k], hi[k], lo[k]);
          asm volatile("" : "+v"(lo[k][0]), "+v"(lo[k][1]));
        } else {
          hi[k] = pack4<TO>(y[k]);
        }
        asm volatile("" : "+v"(hi[k][0]), "+v"(hi[k][1]));  // materialised here: nothing of the packing sinks behind a store
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane * 4 + k * 256;
        if (i < D) {
          TO* op = out + row * (D * kPlanes<TO>) + i;  // split-half rows: [hi: D | lo: D]
          *(i32x2_t*)op = hi[k];
          if constexpr (is_split<TO>::value) *(i32x2_t*)(op + D) = lo[k];
        }
      }
    }
#ifndef MD_LN_NO_PREFETCH
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = vn[k];
    seq = seq_n;
    repl = repl_n;
#endif
  }
}

int launch_layernorm(const float* x, void* out, long rows, int D, float eps, int S, const SeqGroups& g, int prec,
                     int out_f32, hipStream_t s, float fp8_inv_scale, const float* tok0, int tok0_stride, float* x_rw) {
  if (tok0 && x_rw != x) MD_FAIL(MD_ERR_INVALID_ARG, "layernorm: the token-0 replacement writes back into the input rows");
  if (D % 4 != 0 || D > 1024) MD_FAIL(MD_ERR_UNSUPPORTED, "layernorm: D=%d must be a multiple of 4 and <= 1024", D);
  if (rows < 0 || rows > 0x7fffffffL || S <= 0) MD_FAIL(MD_ERR_UNSUPPORTED, "layernorm: %ld rows of sequences of %d", rows, S);
  const int nv = (D + 255) / 256;
  const int grid = grid_for(rows * 64);
  const bool f32o = out_f32 || prec == MD_PREC_F32;
  const bool fp8o = !out_f32 && prec == MD_PREC_FP8;
#define MD_LN(NV)                                                                                                   \
  if (f32o)                                                                                                         \
    hipLaunchKernelGGL((layernorm_kernel<float, NV>), dim3(grid), dim3(256), 0, s, x, (float*)out, rows, D, eps, S, g, 1.f, tok0, tok0_stride, x_rw); \
  else if (fp8o)                                                                                                    \
    hipLaunchKernelGGL((layernorm_kernel<fp8_t, NV>), dim3(grid), dim3(256), 0, s, x, (fp8_t*)out, rows, D, eps, S, g, \
                       fp8_inv_scale, tok0, tok0_stride, x_rw);                                                                              \
  else if (prec == MD_PREC_F16)                                                                                     \
    hipLaunchKernelGGL((layernorm_kernel<f16_t, NV>), dim3(grid), dim3(256), 0, s, x, (f16_t*)out, rows, D, eps, S, g, 1.f, tok0, tok0_stride, x_rw); \
  else if (prec == MD_PREC_F16X2)                                                                                   \
    hipLaunchKernelGGL((layernorm_kernel<f16s_t, NV>), dim3(grid), dim3(256), 0, s, x, (f16s_t*)out, rows, D, eps, S, g, 1.f, tok0, tok0_stride, x_rw); \
  else                                                                                                              \
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, NV>), dim3(grid), dim3(256), 0, s, x, (bf16_t*)out, rows, D, eps, S, g, 1.f, tok0, tok0_stride, x_rw);
  switch (nv) {
    case 1: MD_LN(1) break;
    case 2: MD_LN(2) break;
    case 3: MD_LN(3) break;
    default: MD_LN(4) break;
  }
#undef MD_LN
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// element-type converters
// ------------------------------------------------------------------------------------------------
// `width` = logical row width; split-half tensors store a row as [hi: width | lo: width] (one-plane types ignore it)
template <typename T>
__global__ void f32_to_rows_kernel(const float* __restrict__ in, long n, T* __restrict__ out, int width) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if constexpr (is_split<T>::value) {
      const long row = i / width;
      store1s<T>(out + row * 2 * width + (i - row * width), width, in[i]);
    } else {
      st1<T>(out + i, in[i]);
    }
  }
}
template <typename T>
__global__ void rows_to_f32_kernel(const T* __restrict__ in, long n, float* __restrict__ out, int width) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if constexpr (is_split<T>::value) {
      const long row = i / width;
      out[i] = load1s<T>(in + row * 2 * width + (i - row * width), width);
    } else {
      out[i] = ld1<T>(in + i);
    }
  }
}

int launch_f32_to_rows(const float* in, long count, void* out, int prec, hipStream_t s, int width) {
  if (prec == MD_PREC_F16X2 && (width <= 0 || count % width != 0)) MD_FAIL(MD_ERR_INVALID_ARG, "f32_to_rows: split-half rows need the row width");
  MD_BY_PREC(prec, hipLaunchKernelGGL(f32_to_rows_kernel<T>, dim3(grid_for(count)), dim3(256), 0, s, in, count, (T*)out, width));
  MD_HIP(hipGetLastError());
  return MD_OK;
}
// ---- LayerNorm fold: (mean_t, M2_t) x 4 -> (rstd, -mu rstd) per row ----
__global__ __launch_bounds__(256) void ln_finish_kernel(const float* __restrict__ parts, float* __restrict__ ab, long rows, float inv_n, float eps) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const f32x4_t p0 = *(const f32x4_t*)(parts + r * 8), p1 = *(const f32x4_t*)(parts + r * 8 + 4);
  const float mu = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
  const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
  const float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + 256.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
  const float rstd = 1.0f / sqrtf(m2 * inv_n + eps);
  ab[r * 2] = rstd;
  ab[r * 2 + 1] = -mu * rstd;
}

__global__ __launch_bounds__(256) void fill_pairs_kernel(float* __restrict__ ab, long rows, float a, float b) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r < rows) { ab[2 * r] = a; ab[2 * r + 1] = b; }
}
int launch_fill_pairs(float* ab, long rows, float a, float b, hipStream_t s) {
  if (!ab || rows <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "fill_pairs: invalid argument");
  hipLaunchKernelGGL(fill_pairs_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, ab, rows, a, b);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int launch_ln_finish(const float* parts, float* ab, long rows, float inv_n, float eps, hipStream_t s) {
  if (!parts || !ab || rows <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "ln_finish: invalid argument");
  hipLaunchKernelGGL(ln_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, parts, ab, rows, inv_n, eps);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ---- LayerNorm fold vectors (one wave per output row n) ----
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_vectors_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias, int N, int K,
                                                              float* __restrict__ c, float* __restrict__ d) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  const float* w = W + (long)n * K;
  double sc = 0.0, sd = 0.0;
  for (int k = lane; k < K; k += 64) {
    float wr;
    if constexpr (sizeof(T) == 4) {
      wr = w[k];
    } else {
      const float hi = (float)cvt_elem<T>(w[k]);
      wr = is_split<T>::value ? hi + (float)cvt_elem<T>(w[k] - hi) : hi;
    }
    sc += (double)gamma[k] * (double)wr;
    sd += (double)beta[k] * (double)wr;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sc += __shfl_xor(sc, o);
    sd += __shfl_xor(sd, o);
  }
  if (lane == 0) {
    c[n] = (float)sc;
    d[n] = (float)(sd + (bias ? (double)bias[n] : 0.0));
  }
}

int launch_ln_fold_vectors(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, int prec, float* c,
                           float* d, hipStream_t s) {
  if (!W || !gamma || !beta || !c || !d || N <= 0 || K <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "ln_fold_vectors: invalid argument");
  MD_BY_PREC(prec, hipLaunchKernelGGL(ln_fold_vectors_kernel<T>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, gamma, beta, bias, N, K, c, d));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

int launch_convert_rows(const float* x, void* out, long count, int prec, hipStream_t s, int width) {
  return launch_f32_to_rows(x, count, out, prec, s, width);
}
int launch_rows_to_f32(const void* in, long count, float* out, int prec, hipStream_t s, int width) {
  if (prec == MD_PREC_F16X2 && (width <= 0 || count % width != 0)) MD_FAIL(MD_ERR_INVALID_ARG, "rows_to_f32: split-half rows need the row width");
  MD_BY_PREC(prec, hipLaunchKernelGGL(rows_to_f32_kernel<T>, dim3(grid_for(count)), dim3(256), 0, s, (const T*)in, count, out, width));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int B, int C, int H, int W, T* __restrict__ out,
                                    int relu, long ld) {
  const long total = (long)B * C * H * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    long t = e / C;
    const int xw = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float v = in[(((long)b * C + c) * H + y) * W + xw];
    if (relu) v = fmaxf(v, 0.f);
    st1p<T>(out + (e / C) * (ld * kPlanes<T>) + c, ld, v);  // ld = logical pixel stride; split-half pixels: [hi: ld | lo: ld]
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, int B, int C, int H, int W, long ld, int coff,
                                    float* __restrict__ out) {
  const long total = (long)B * C * H * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(e % W);
    long t = e / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const long src = (((long)b * H + y) * W + xw) * (ld * kPlanes<T>) + coff + c;  // ld = logical row width (split-half: [hi: ld | lo: ld])
    out[e] = ld1p<T>(in + src, ld);
  }
}

int launch_nchw_to_nhwc(const float* in, int B, int C, int H, int W, void* out, int prec, int relu, hipStream_t s, long ld) {
  const long total = (long)B * C * H * W;
  if (ld != 0 && ld < C) MD_FAIL(MD_ERR_INVALID_ARG, "nchw_to_nhwc: pixel stride %ld below C %d", ld, C);
  MD_BY_PREC(prec, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, in, B, C, H, W, (T*)out, relu, ld ? ld : (long)C));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// `head_debug`'s un-fused tail (depth_pro/mod.rs:293-296): pre_out = conv_out 1x1 (C -> 1, + bias) on the materialised relu(conv1) map,
// canonical = relu(pre_out). One thread per pixel (a debug entry: the product path does this inside the fused head epilogue).
template <typename T>
__global__ void head_tail_debug_kernel(const T* __restrict__ relu_map, long ld, long pixels, int C, const float* __restrict__ w,
                                       const float* __restrict__ bias, float* __restrict__ pre_out, float* __restrict__ canonical) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < pixels; e += (long)gridDim.x * blockDim.x) {
    const T* px = relu_map + e * (ld * kPlanes<T>);
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(ld1p<T>(px + c, ld), w[c], acc);
    acc += bias ? bias[0] : 0.f;
    if (pre_out) pre_out[e] = acc;
    if (canonical) canonical[e] = fmaxf(acc, 0.f);
  }
}
int launch_head_tail_debug(const void* relu_map, long ld, long pixels, int C, const float* w, const float* bias, float* pre_out,
                           float* canonical, int prec, hipStream_t s) {
  MD_BY_PREC(prec, hipLaunchKernelGGL(head_tail_debug_kernel<T>, dim3(grid_for(pixels)), dim3(256), 0, s, (const T*)relu_map, ld, pixels, C, w,
                                      bias, pre_out, canonical));
  MD_HIP(hipGetLastError());
  return MD_OK;
}
int launch_nhwc_to_nchw(const void* in, int B, int C, int H, int W, long ld, int coff, float* out, int prec,
                        hipStream_t s) {
  const long total = (long)B * C * H * W;
  MD_BY_PREC(prec, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)in, B, C, H, W, ld, coff, out));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// Small direct convolution (FOV head), fp32 arithmetic.  One wave per (output pixel, CO output
// channels): lanes stride over the k*k*Cin window in 4-channel vectors, the input window is read
// once and reused for the CO weight rows.
// ------------------------------------------------------------------------------------------------
template <typename TI>
__device__ inline void load4v(const TI* in, long src, long plane, float v[4]) {
  const f32x4_t t = load4p<TI>(in + src, plane);
  v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
}

template <typename TI, int CO, int VEC>
__global__ __launch_bounds__(256) void conv_direct_kernel(const TI* __restrict__ in, const float* __restrict__ add,
                                                          int B, int H, int W, int Cin, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int Cout, int k, int stride,
                                                          int pad, int relu, float* __restrict__ out, int OH, int OW, int out_ld) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int cgroups = (Cout + CO - 1) / CO;
  const long total = (long)B * OH * OW * cgroups;
  const int KK = k * k * Cin;
  for (long e = wave_id; e < total; e += nwaves) {
    const int co0 = (int)(e % cgroups) * CO;
    long t = e / cgroups;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int idx = lane * VEC; idx < KK; idx += 64 * VEC) {
      const int ci = idx % Cin;
      const int tap = idx / Cin;
      const int ky = tap / k, kx = tap % k;
      const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const long src = (((long)b * H + iy) * W + ix) * Cin + ci;                       // fp32 addend: one plane
      const long srci = (((long)b * H + iy) * W + ix) * (Cin * kPlanes<TI>) + ci;      // input pixel: [hi: Cin | lo: Cin] when split
      float v[4];
      if constexpr (VEC == 4) {
        load4v<TI>(in, srci, Cin, v);
        if (add) {
          const float4 a4 = *(const float4*)(add + src);
          v[0] += a4.x, v[1] += a4.y, v[2] += a4.z, v[3] += a4.w;
        }
      } else {
        v[0] = ld1p<TI>(in + srci, Cin);
        if (add) v[0] += add[src];
      }
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        if (co0 + c < Cout) {
          const float* wr = w + (long)(co0 + c) * KK + idx;
          if constexpr (VEC == 4) {
            const float4 w4 = *(const float4*)wr;
            acc[c] += v[0] * w4.x + v[1] * w4.y + v[2] * w4.z + v[3] * w4.w;
          } else {
            acc[c] += v[0] * wr[0];
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float sum = wave_sum(acc[c]);
      if (lane == 0 && co0 + c < Cout) {
        float r = sum + (bias ? bias[co0 + c] : 0.f);
        if (relu) r = fmaxf(r, 0.f);
        out[(((long)b * OH + oy) * OW + ox) * out_ld + co0 + c] = r;
      }
    }
  }
}

template <typename TI>
static void conv_direct_dispatch(const TI* in, const float* add, int B, int H, int W, int Cin, const float* w,
                                 const float* bias, int Cout, int k, int stride, int pad, int relu, float* out, int OH,
                                 int OW, int out_ld, hipStream_t s) {
  const bool vec = Cin % 4 == 0;
  const int co = Cout >= 8 ? 8 : 1;
  const long total = (long)B * OH * OW * ((Cout + co - 1) / co);
  const int grid = grid_for(total * 64);
#define MD_CD(CO, VEC)                                                                                               \
  hipLaunchKernelGGL((conv_direct_kernel<TI, CO, VEC>), dim3(grid), dim3(256), 0, s, in, add, B, H, W, Cin, w, bias, \
                     Cout, k, stride, pad, relu, out, OH, OW, out_ld)
  if (co == 8 && vec)
    MD_CD(8, 4);
  else if (co == 8)
    MD_CD(8, 1);
  else if (vec)
    MD_CD(1, 4);
  else
    MD_CD(1, 1);
#undef MD_CD
}

int launch_conv_direct(const void* in, int in_prec, const float* add, int B, int H, int W, int Cin, const float* w,
                       const float* bias, int Cout, int k, int stride, int pad, int relu, float* out, hipStream_t s, int out_ld) {
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  if (OH <= 0 || OW <= 0) MD_FAIL(MD_ERR_SHAPE, "conv_direct: input %dx%d smaller than kernel %d", H, W, k);
  if (out_ld != 0 && out_ld < Cout) MD_FAIL(MD_ERR_INVALID_ARG, "conv_direct: output pixel stride %d below Cout %d", out_ld, Cout);
  MD_BY_PREC(in_prec, conv_direct_dispatch<T>((const T*)in, add, B, H, W, Cin, w, bias, Cout, k, stride, pad, relu, out, OH, OW, out_ld ? out_ld : Cout, s));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) operand preparation
// ------------------------------------------------------------------------------------------------
__global__ void f32_to_fp8_kernel(const float* __restrict__ in, long n4, float inv_scale, int* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4_t v = *(const f32x4_t*)(in + 4 * i) * inv_scale;
    out[i] = pack4_fp8(v[0], v[1], v[2], v[3]);
  }
}

int launch_f32_to_fp8(const float* in, long n, float inv_scale, void* out, hipStream_t s) {
  if (n % 4 != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "f32_to_fp8: count must be a multiple of 4");
  hipLaunchKernelGGL(f32_to_fp8_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, in, n / 4, inv_scale, (int*)out);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

__global__ __launch_bounds__(256) void pack_fp8_rows_kernel(const float* __restrict__ w, int N, int K, int Kp,
                                                            int* __restrict__ out, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long n = wave_id; n < N; n += nwaves) {
    const float* row = w + n * K;
    float amax = 0.f;
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(row[k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[n] = sc;
    for (int k4 = lane; k4 < Kp / 4; k4 += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (4 * k4 + j) < K ? row[4 * k4 + j] * inv : 0.f;
      out[n * (Kp / 4) + k4] = pack4_fp8(v[0], v[1], v[2], v[3]);
    }
  }
}

int launch_pack_fp8_rows(const float* w, int N, int K, int Kp, void* out, float* scale, hipStream_t s) {
  if (Kp % 4 != 0 || Kp < K) MD_FAIL(MD_ERR_INVALID_ARG, "pack_fp8_rows: bad padded K");
  hipLaunchKernelGGL(pack_fp8_rows_kernel, dim3(grid_for((long)N * 64)), dim3(256), 0, s, w, N, K, Kp, (int*)out, scale);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// Depth-Anything-v3 `small` backbone extras (restated from the public DA3 definition; oracle/da3_ref.py)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(T* __restrict__ qk, long rows, int S, int NT, int D, int heads,
                                                           int pw, const float* __restrict__ qg,
                                                           const float* __restrict__ qb, const float* __restrict__ kg,
                                                           const float* __restrict__ kb, float eps,
                                                           const float* __restrict__ rope_cos,
                                                           const float* __restrict__ rope_sin, int global_pos, float q_scale) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const long total = rows * heads * 2;
  for (long e = wave_id; e < total; e += nwaves) {
    const int which = (int)(e & 1);
    const int hd = (int)((e >> 1) % heads);
    const long row = (e >> 1) / heads;
    const int t = (int)(row % S);
    if (t >= NT) continue;  // wave-uniform
    T* p = qk + (row * 2L * D + (long)which * D) * kPlanes<T> + hd * 64 + lane;  // split-half rows: [q_hi | q_lo | k_hi | k_lo]
    const float v = ld1p<T>(p, D);
    const float mean = wave_sum(v) * (1.0f / 64.0f);
    const float c = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(c * c) * (1.0f / 64.0f) + eps);
    const float y = c * rstd * (which ? kg[lane] : qg[lane]) + (which ? kb[lane] : qb[lane]);
    int py = 0, px = 0;
    if (t > 0) {
      if (global_pos) {
        py = px = 1;
      } else {
        const int pi = t - 1;
        py = pi / pw;
        px = pi - py * pw + 1;
        py += 1;
      }
    }
    const int pos = (lane >> 5) ? px : py;  // first half of head_dim rotates with the row, second with the column
    const int jj = lane & 31, f = jj & 15;
    const float cs = rope_cos[pos * 16 + f], sn = rope_sin[pos * 16 + f];
    const float partner = __shfl_xor(y, 16);
    float o = jj < 16 ? y * cs - partner * sn : y * cs + partner * sn;
    if (!which) o *= q_scale;  // the softmax scale rides on q (see GemmParams::qscale); the LayerNorm above removed the epilogue's
    st1p<T>(p, D, o);
  }
}

int launch_qk_norm_rope(void* qk, long rows, int S, int n_tokens, int D, int heads, int pw, const float* q_gamma,
                        const float* q_beta, const float* k_gamma, const float* k_beta, float eps, const float* rope_cos,
                        const float* rope_sin, int global_pos, float q_scale, int prec, hipStream_t s) {
  if (D != heads * 64) MD_FAIL(MD_ERR_UNSUPPORTED, "qk_norm_rope: head_dim must be 64");
  const int grid = grid_for(rows * heads * 2 * 64);
  MD_BY_PREC(prec, hipLaunchKernelGGL(qk_norm_rope_kernel<T>, dim3(grid), dim3(256), 0, s, (T*)qk, rows, S, n_tokens, D, heads, pw, q_gamma, q_beta, k_gamma, k_beta, eps, rope_cos, rope_sin, global_pos, q_scale));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

__global__ void set_token0_kernel(float* __restrict__ x, int nseq, int S, int D, const float* __restrict__ src, int src_stride) {
  const long total = (long)nseq * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / D), d = (int)(i - (long)b * D);
    x[(long)b * S * D + d] = src[(long)b * src_stride + d];
  }
}

int launch_set_token0(float* x, int nseq, int S, int D, const float* src, hipStream_t s, int src_stride) {
  hipLaunchKernelGGL(set_token0_kernel, dim3(grid_for((long)nseq * D)), dim3(256), 0, s, x, nseq, S, D, src, src_stride);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

template <typename TO>
__global__ __launch_bounds__(256) void hook_cat_ln_kernel(const float* __restrict__ xl, const float* __restrict__ x,
                                                          long rows, int S, int NT, int D, const float* __restrict__ ng,
                                                          const float* __restrict__ nb, float eps_final,
                                                          const float* __restrict__ hg, const float* __restrict__ hb,
                                                          float eps_head, TO* __restrict__ out,
                                                          float* __restrict__ cam_out) {
  constexpr int MAXV = 16;  // D <= 1024
  const int lane = threadIdx.x & 63;
  const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int nv = D >> 6;
  for (long row = wave_id; row < rows; row += nwaves) {
    const int t = (int)(row % S);
    if (t >= NT) continue;
    float a[MAXV], b[MAXV];
    float sb = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
      if (k < nv) {
        a[k] = xl[row * D + lane + 64 * k];
        b[k] = x[row * D + lane + 64 * k];
        sb += b[k];
      }
    if (cam_out && t == 0) {
      const long sq = row / S;
#pragma unroll
      for (int k = 0; k < MAXV; ++k)
        if (k < nv) {
          cam_out[sq * 2 * D + lane + 64 * k] = a[k];
          cam_out[sq * 2 * D + D + lane + 64 * k] = b[k];
        }
    }
    const float mb = wave_sum(sb) / (float)D;
    float qb = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
      if (k < nv) qb += (b[k] - mb) * (b[k] - mb);
    const float rb = 1.0f / sqrtf(wave_sum(qb) / (float)D + eps_final);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
      if (k < nv) {
        b[k] = (b[k] - mb) * rb * ng[lane + 64 * k] + nb[lane + 64 * k];
        sum += a[k] + b[k];
      }
    const float mean = wave_sum(sum) / (float)(2 * D);
    float sq2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
      if (k < nv) sq2 += (a[k] - mean) * (a[k] - mean) + (b[k] - mean) * (b[k] - mean);
    const float rstd = 1.0f / sqrtf(wave_sum(sq2) / (float)(2 * D) + eps_head);
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
      if (k < nv) {
        const int i = lane + 64 * k;
        const float y1 = (a[k] - mean) * rstd * hg[i] + hb[i];
        const float y2 = (b[k] - mean) * rstd * hg[D + i] + hb[D + i];
        TO* orow = out + row * (2L * D * kPlanes<TO>);  // split-half rows: [hi: 2D | lo: 2D]
        st1p<TO>(orow + i, 2L * D, y1);
        st1p<TO>(orow + D + i, 2L * D, y2);
      }
  }
}

int launch_hook_cat_ln(const float* x_local, const float* x, long rows, int S, int n_tokens, int D, const float* norm_g,
                       const float* norm_b, float eps_final, const float* head_g, const float* head_b, float eps_head,
                       void* out, float* cam_out, int prec, hipStream_t s) {
  if (D % 64 != 0 || D > 1024) MD_FAIL(MD_ERR_UNSUPPORTED, "hook_cat_ln: D=%d must be a multiple of 64 and <= 1024", D);
  const int grid = grid_for(rows * 64);
  MD_BY_PREC(prec, hipLaunchKernelGGL(hook_cat_ln_kernel<T>, dim3(grid), dim3(256), 0, s, x_local, x, rows, S, n_tokens, D, norm_g, norm_b, eps_final, head_g, head_b, eps_head, (T*)out, cam_out));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

__global__ void pose_to_camera_kernel(const float* __restrict__ pose, int B, int H, int W, float* __restrict__ extr,
                                      float* __restrict__ intr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  pose_to_camera_one(pose + b * 9, H, W, extr ? extr + b * 12 : nullptr, intr ? intr + b * 9 : nullptr);
}

int launch_pose_to_camera(const float* pose, int B, int H, int W, float* extrinsics, float* intrinsics, hipStream_t s) {
  hipLaunchKernelGGL(pose_to_camera_kernel, dim3((B + 63) / 64), dim3(64), 0, s, pose, B, H, W, extrinsics, intrinsics);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// a12/a13 scalar tail (depth_pro/mod.rs:330-346, 370-414)
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline void fov_scalar_math(float fovx_deg, int H, int W, float* focal_px, float* fovy_rad,
                                                float* ratio) {
#pragma clang fp contract(off)
  const float fovx_rad = fovx_deg * (float)(3.14159265358979323846 / 180.0);
  const float denom = tanf(fovx_rad * 0.5f);
  const float focal = ((float)W * 0.5f) / denom;
  if (focal_px) *focal_px = focal;
  if (ratio) *ratio = (float)W / focal;
  if (fovy_rad) {
    const float k = (float)0.273;
    const float pi4 = (float)0.78539816339744830962, pi2 = (float)1.57079632679489661923;
    const float aspect = (float)((double)H / (double)W);
    const float t = tanf(fovx_rad * 0.5f) * aspect;
    const float sgn = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
    const float ax = fabsf(t);
    const float use_inv = ax > 1.0f ? 1.f : 0.f;
    const float inv = 1.0f / ax;
    const float xr = ax * (1.0f - use_inv) + (use_inv != 0.f ? inv * use_inv : 0.f);
    const float inner = (1.0f - xr) * k + pi4;
    const float atan_reduced = xr * inner;
    const float delta = pi2 - atan_reduced * 2.0f;
    const float atan_ax = atan_reduced + delta * use_inv;
    *fovy_rad = atan_ax * sgn * 2.0f;
  }
}

__global__ void fov_post_kernel(const float* __restrict__ fov_deg, int B, int H, int W, float* focal_px,
                                float* fovy_rad, float* ratio) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) fov_scalar_math(fov_deg[b], H, W, focal_px ? focal_px + b : nullptr, fovy_rad ? fovy_rad + b : nullptr,
                             ratio ? ratio + b : nullptr);
}

int launch_fov_post(const float* fov_deg, int B, int H, int W, float* focal_px, float* fovy_rad, float* ratio,
                    hipStream_t s) {
  hipLaunchKernelGGL(fov_post_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, fov_deg, B, H, W, focal_px, fovy_rad, ratio);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

void fov_scalar_host(float fovx_deg, int H, int W, float* focal_px, float* fovy_rad) {
  fov_scalar_math(fovx_deg, H, W, focal_px, fovy_rad, nullptr);
}

__global__ void depth_post_kernel(const float* __restrict__ canonical, const float* __restrict__ ratio, int B, long hw,
                                  float* __restrict__ out, int post) {
#pragma clang fp contract(off)
  const long total = (long)B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = canonical[i] * ratio[i / hw];
    if (post == 1) v = 1.0f / fminf(fmaxf(v, 1e-4f), 1e4f);
    out[i] = v;
  }
}

int launch_depth_post(const float* canonical, const float* ratio, int B, long hw, float* out, int post, hipStream_t s) {
  hipLaunchKernelGGL(depth_post_kernel, dim3(grid_for((long)B * hw)), dim3(256), 0, s, canonical, ratio, B, hw, out,
                     post);
  MD_HIP(hipGetLastError());
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// row softmax on fp32 scores (fp32 attention path): one wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, long rows, int n_valid, int ld,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long row = wave_id; row < rows; row += nwaves) {
    float* r = s + row * ld;
    float mx = -INFINITY;
    for (int i = lane; i < n_valid; i += 64) mx = fmaxf(mx, r[i] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < n_valid; i += 64) {
      const float e = expf(r[i] * scale - mx);
      r[i] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < ld; i += 64) r[i] = i < n_valid ? r[i] * inv : 0.f;
  }
}

int launch_softmax_rows(float* s, long rows, int n_valid, int ld, float scale, hipStream_t st) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, st, s, rows, n_valid, ld, scale);
  MD_HIP(hipGetLastError());
  return MD_OK;
}


// ------------------------------------------------------------------------------------------------
// stand-alone attention op helpers
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void qkv_split_kernel(const float* __restrict__ qkv, int Tn, int N, int heads, int SS, int kpad,
                                 T* __restrict__ qk, T* __restrict__ vT, float q_scale) {
  const int D = heads * 64;
  const long total = (long)Tn * N * 3 * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % (3 * D));
    const long t = e / (3 * D);
    const int i = (int)(t % N);
    const int seq = (int)(t / N);
    const float v = c < D ? qkv[e] * q_scale : qkv[e];
    if (c < 2 * D) {  // split-half rows: [q_hi | q_lo | k_hi | k_lo], each D wide
      const int isk = c >= D ? 1 : 0;
      st1p<T>(qk + ((long)seq * SS + i) * (2 * D * kPlanes<T>) + isk * (D * kPlanes<T>) + (c - isk * D), D, v);
    } else {          // split-half V^T: the lo plane behind the whole hi plane
      const int cc = c - 2 * D;
      st1p<T>(vT + (((long)seq * heads + (cc >> 6)) * 64 + (cc & 63)) * kpad + i, (long)Tn * heads * 64 * kpad, v);
    }
  }
}

int launch_qkv_split(const float* qkv, int Tn, int N, int heads, int SS, int kpad, void* qk, void* vT, float q_scale, int prec,
                     hipStream_t s) {
  const long total = (long)Tn * N * 3 * heads * 64;
  MD_BY_PREC(prec, hipLaunchKernelGGL(qkv_split_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, qkv, Tn, N, heads, SS, kpad, (T*)qk, (T*)vT, q_scale));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

template <typename T>
__global__ void unpad_rows_kernel(const T* __restrict__ in, int Tn, int N, int SS, int D, float* __restrict__ out) {
  const long total = (long)Tn * N * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const long t = e / D;
    const int i = (int)(t % N);
    const int seq = (int)(t / N);
    const long src = ((long)seq * SS + i) * (D * kPlanes<T>) + d;
    out[e] = ld1p<T>(in + src, D);
  }
}

int launch_unpad_rows(const void* in, int Tn, int N, int SS, int D, float* out, int prec, hipStream_t s) {
  const long total = (long)Tn * N * D;
  MD_BY_PREC(prec, hipLaunchKernelGGL(unpad_rows_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)in, Tn, N, SS, D, out));
  MD_HIP(hipGetLastError());
  return MD_OK;
}

}  // namespace md
