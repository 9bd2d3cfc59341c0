#include "gemm_impl.h"

namespace md {
int launch_gemm_f32(GemmParams& p, int amode, int tile, hipStream_t stream) {
  return launch_gemm_typed<float>(p, amode, tile, stream);
}
}  // namespace md
