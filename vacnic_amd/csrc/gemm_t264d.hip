// one tile configuration of the MFMA GEMM (see gemm_kernel.h / gemm.hip): 256x128 ping-pong + activation-dropout epilogues
#include "gemm_kernel.h"
namespace vacgemm {
int launch_t264d(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s) { return launch_gemm<256, 128, 2, 4, 32, 4, true, false, true>(p, xks, wks, zsplits, s); }
}
