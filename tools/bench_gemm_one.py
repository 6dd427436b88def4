"""One GEMM shape, a few launches, for PMC collection: python tools/bench_gemm_one.py M N K hint [xks wks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
M, N, Kd, hint = (int(v) for v in sys.argv[1:5])
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
x = r(M, Kd); w = r(N, Kd); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(8):
    K.gemm(x, w, M, N, Kd, out=out, tile_hint=hint)
torch.cuda.synchronize()
