// pose_encoding_to_extri_intri for one camera (camera.rs:281-358, quaternion_to_matrix :360-416): shared by the stand-alone
// pose_to_camera kernel (ops.hip) and the camera decoder's tail (camera.hip) so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace md {

// p = (t3 | quat xyzw | fov_h fov_w); extr [3][4] world-to-camera = [R^T | -R^T t], intr [3][3]; either output may be null
__device__ inline void pose_to_camera_one(const float* __restrict__ p, int H, int W, float* __restrict__ e, float* __restrict__ k) {
#pragma clang fp contract(off)
  const float tx = p[0], ty = p[1], tz = p[2], x = p[3], y = p[4], z = p[5], w = p[6], fh = p[7], fw = p[8];
  // quaternion_to_matrix, no normalisation, as in the reference
  float R[3][3];
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - w * z); R[0][2] = 2.f * (x * z + w * y);
  R[1][0] = 2.f * (x * y + w * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - w * x);
  R[2][0] = 2.f * (x * z - w * y); R[2][1] = 2.f * (y * z + w * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
  if (e) {
    for (int i = 0; i < 3; ++i) {
      const float r0 = R[0][i], r1 = R[1][i], r2 = R[2][i];
      e[i * 4 + 0] = r0; e[i * 4 + 1] = r1; e[i * 4 + 2] = r2;
      e[i * 4 + 3] = -(r0 * tx + r1 * ty + r2 * tz);
    }
  }
  if (k) {
    const float th = sinf(fh * 0.5f) / cosf(fh * 0.5f), tw = sinf(fw * 0.5f) / cosf(fw * 0.5f);
    const float hh = (float)H / 2.0f, wh = (float)W / 2.0f;
    k[0] = wh / tw; k[1] = 0.f; k[2] = wh;
    k[3] = 0.f; k[4] = hh / th; k[5] = hh;
    k[6] = 0.f; k[7] = 0.f; k[8] = 1.f;
  }
}

}  // namespace md
