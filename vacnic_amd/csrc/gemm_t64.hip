// one tile configuration of the MFMA GEMM (gemm_kernel.h) per translation unit: <BM, BN, WM, WN, BKT, NSTAGE, PIPE> = <64, 128, 2, 2, 64, 4, false>
#include "gemm_kernel.h"
namespace vacgemm {
int launch_t64(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s) { return launch_gemm<64, 128, 2, 2, 64, 4, false>(p, xks, wks, zsplits, s); }
}  // namespace vacgemm
