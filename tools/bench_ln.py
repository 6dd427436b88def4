"""add_ln fwd/bwd variants under rocprofv3 (tools/kt.sh): isolates the dgamma/dbeta atomics and the dropout recompute."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
R, D = 16384, 1024
dev = "cuda"
x = torch.randn(R, D, device=dev).bfloat16(); rs = torch.randn(R, D, device=dev).bfloat16(); dy = torch.randn(R, D, device=dev).bfloat16()
g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
out, mean, rstd = K.add_ln_fwd(x, rs, g, b, p_drop=0.1, seed=5)[:3]
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
sep = torch.zeros(1024, device=dev)
cases = [("p=0.1 dgamma", dict(p_drop=0.1, seed=5), dg, db), ("p=0.1 no dgamma", dict(p_drop=0.1, seed=5), None, None),
         ("p=0 dgamma", dict(), dg, db), ("p=0 no dgamma", dict(), None, None)]
for name, kw, a, c in cases:
    sep.add_(1.0)
    for _ in range(20):
        K.add_ln_bwd(dy, x, rs, g, mean, rstd, a, c, **kw)
    print(name)
sep.add_(1.0)
for _ in range(20):
    K.add_ln_fwd(x, rs, g, b, p_drop=0.1, seed=5)
sep.add_(1.0)
for _ in range(20):
    K.add_ln_fwd(x, rs, g, b)
torch.cuda.synchronize()
