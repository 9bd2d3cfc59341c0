#include "gemm_impl.h"

namespace md {
int launch_gemm_bf16(GemmParams& p, int amode, int tile, hipStream_t stream) {
  return launch_gemm_typed<bf16_t>(p, amode, tile, stream);
}
}  // namespace md
