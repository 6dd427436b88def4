// grouped weight-gradient GEMM (gemm_kernel.h: gemm_group_kernel), 128x128 tiles, software-pipelined loop, both operands K-strided
#include "gemm_kernel.h"
namespace vacgemm {
int launch_group128(const GroupP& g, hipStream_t s) { return launch_gemm_group<128, 128, 2, 2, 64, 2, true>(g, s); }
}  // namespace vacgemm
