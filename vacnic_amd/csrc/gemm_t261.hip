// one tile configuration of the MFMA GEMM (gemm_kernel.h) per translation unit: <BM, BN, WM, WN, BKT, NSTAGE, PIPE> = <128, 128, 2, 2, 64, 2, false>
#include "gemm_kernel.h"
namespace vacgemm {
int launch_t261(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s) { return launch_gemm<128, 128, 2, 2, 64, 2, false>(p, xks, wks, zsplits, s); }
}  // namespace vacgemm
