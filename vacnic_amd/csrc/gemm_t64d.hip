// one tile configuration of the MFMA GEMM (see gemm_kernel.h / gemm.hip): 64x128, 4-deep ring + activation-dropout epilogues
#include "gemm_kernel.h"
namespace vacgemm {
int launch_t64d(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s) { return launch_gemm<64, 128, 2, 2, 64, 4, false, false, true>(p, xks, wks, zsplits, s); }
}
