// fp8 (OCP e4m3) operand instantiation of the MFMA GEMM family (outputs / residuals stay bf16 or f32).
#include "gemm_impl.h"

namespace md {
int launch_gemm_fp8(GemmParams& p, int amode, int tile, hipStream_t stream) {
  if (amode != A_DENSE) MD_FAIL(MD_ERR_UNSUPPORTED, "fp8 operands are built for dense GEMMs only");
  return launch_tile<fp8_t, A_DENSE>(p, tile, stream);
}
}  // namespace md
