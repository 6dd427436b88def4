"""Same attention work, two memory layouts: heads interleaved in a [B,T,3d] buffer (the model's) vs one head per batch row
(contiguous 128-byte rows).  Run under tools/kt.sh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
sep = torch.zeros(1024, device="cuda")
B, H, T = 32, 16, 512
d = H * 64
qkv = r(B, T, 3 * d)
sep.add_(1.0)
for _ in range(10):
    K.attn_fwd(qkv[..., 2 * d:], qkv[..., :d], qkv[..., d:2 * d], B, H, T, T, need_lse=False)
print("interleaved heads, row stride", 3 * d)
q1 = r(B * H, T, 64); k1 = r(B * H, T, 64); v1 = r(B * H, T, 64)
sep.add_(1.0)
for _ in range(10):
    K.attn_fwd(q1, k1, v1, B * H, 1, T, T, need_lse=False)
print("one head per batch row, row stride 64")
sep.add_(1.0)
torch.cuda.synchronize()
