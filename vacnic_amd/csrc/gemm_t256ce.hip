// the 256x256 ping-pong configuration with the fused LM-head cross-entropy epilogues (out_mode 3 / 4), forward layout only
#include "gemm_kernel.h"
namespace vacgemm {
int launch_t256ce(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s) { return launch_gemm<256, 256, 2, 4, 32, 4, true, true>(p, xks, wks, zsplits, s); }
}  // namespace vacgemm
