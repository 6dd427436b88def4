"""wgrad GEMM (TT, split-K atomics) cost split: python tools/bench_wgrad.py  (under tools/kt.sh, CHUNK=1)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
sep = torch.zeros(1024, device="cuda")
M, N, Kd = 1024, 1024, 16384          # dW[N_out=1024, K_in=1024] += dY^T[.,16384] X
dy = r(Kd, M); x = r(Kd, N); out = torch.zeros(M, N, device="cuda")
CASES = [(128, 8), (264, 8), (4264, 8), (264, 4), (264, 16), (128, 8), (264, 8)]
for hint, split in CASES:
    sep.add_(1.0)
    for _ in range(20):
        K.gemm(dy, x, M, N, Kd, out=out, ldx=M, ldw=N, ldo=N, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=split, tile_hint=hint)
    print(hint, split)
sep.add_(1.0)
torch.cuda.synchronize()
