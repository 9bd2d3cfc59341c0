// IEEE half operand instantiation of the MFMA GEMM family (v_mfma_f32_*_f16; outputs / residuals f16 or f32).
#include "gemm_impl.h"

namespace md {
int launch_gemm_f16(GemmParams& p, int amode, int tile, hipStream_t stream) {
  return launch_gemm_typed<f16_t>(p, amode, tile, stream);
}
}  // namespace md
