// Split-half instantiation of the MFMA GEMM family (MD_PREC_F16X2): operands are IEEE-half planes (v_mfma_f32_*_f16),
// an activation row is [hi | lo] and a weight row [W | W] or [Wh | Wh | Wl] -- the main loop is the f16 loop over a
// 2x / 3x longer K; the epilogues write outputs as hi + lo planes.
#include "gemm_impl.h"

namespace md {
int launch_gemm_f16x2(GemmParams& p, int amode, int tile, hipStream_t stream) {
  return launch_gemm_typed<f16s_t>(p, amode, tile, stream);
}
}  // namespace md
