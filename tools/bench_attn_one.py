"""One attention shape, a few launches (PMC collection): python tools/bench_attn_one.py B H Tq Tk [bwd]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
B, H, Tq, Tk = (int(v) for v in sys.argv[1:5])
d = H * 64
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
qkv = r(B, Tq, 3 * d) if Tq == Tk else None
q = qkv[..., 2 * d:] if qkv is not None else r(B, Tq, d)
kv = qkv[..., :2 * d] if qkv is not None else r(B, Tk, 2 * d)
for _ in range(8):
    K.attn_fwd(q, kv[..., :d], kv[..., d:], B, H, Tq, Tk, need_lse=True)
torch.cuda.synchronize()
