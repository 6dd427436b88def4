// one tile configuration of the MFMA GEMM (gemm_kernel.h) per translation unit: <BM, BN, WM, WN, BKT, NSTAGE, PIPE> = <256, 256, 2, 4, 32, 4, true>
#include "gemm_kernel.h"
namespace vacgemm {
int launch_t256(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s) { return launch_gemm<256, 256, 2, 4, 32, 4, true>(p, xks, wks, zsplits, s); }
}  // namespace vacgemm
