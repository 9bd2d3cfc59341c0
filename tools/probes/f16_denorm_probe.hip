// Hardware probe (gfx950): do the f16 MFMAs and the f32 -> f16 conversion keep IEEE-half SUBNORMAL values?
// The hi/lo split of the accurate fast mode (MD_PREC_F16X2) stores lo = f16(x - f16(x)), which is subnormal for
// |x| < 0.25; a flush to zero anywhere would silently turn the mode back into plain f16 for small activations.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/build/f16_denorm_probe tools/probes/f16_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

__global__ void probe(float a, float b, float* out) {
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;  // v_cvt_f16_f32
  h8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = ha; vb[i] = hb; }
  f4 c4 = {0.f, 0.f, 0.f, 0.f};
  c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, c4, 0, 0, 0);
  f16v c16 = {0.f};
  c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c16, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = (float)ha;   // the converted value, widened back
    out[1] = c4[0];       // 32 * a * b
    out[2] = c16[0];      // 16 * a * b
    out[3] = (float)ha * (float)hb;
  }
}

int main() {
  float* d;
  hipMalloc(&d, 64);
  const float as[] = {1.0f, 6.2e-5f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 2.0e-8f};
  int bad = 0;
  for (float a : as) {
    const float b = 1024.0f;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    float h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const float want16 = 32.f * h[0] * b, want32 = 16.f * h[0] * b;
    const bool ok = h[1] == want16 && h[2] == want32 && (a < 3e-8f || h[0] != 0.f);
    printf("a=%.3e  f16(a)=%.6e  mfma16x16x32=%.6e (want %.6e)  mfma32x32x16=%.6e (want %.6e)  %s\n", a, h[0], h[1], want16, h[2], want32,
           ok ? "ok" : "FLUSHED/MISMATCH");
    bad += !ok;
  }
  printf(bad ? "RESULT: f16 subnormals are NOT preserved\n" : "RESULT: f16 subnormals preserved by conversion and MFMA\n");
  return bad ? 1 : 0;
}
