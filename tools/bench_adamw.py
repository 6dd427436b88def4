"""AdamW kernel alone on a BART-large-sized arena (HBM-bound: 16 B read + 14 B written per parameter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
n = 420 * 1024 * 1024
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3
m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); p16 = torch.empty(n, device="cuda", dtype=torch.bfloat16)
hyper = torch.tensor([3e-5, 1.0], device="cuda")
for clip in (None, torch.ones(2, device="cuda")):
    for _ in range(3):
        K.adamw(p, g, m, v, p16, hyper, n, zero_grad=False, clip_coef=clip)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.adamw(p, g, m, v, p16, hyper, n, zero_grad=True, clip_coef=clip)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"clip={'on' if clip is not None else 'off'}: {ms:.3f} ms for {n/1e6:.0f}M params = {n*34/ms/1e9:.2f} TB/s (34 B/param incl. grad zeroing)")
out = K.grad_clip_coef(g, n, 0.1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    K.grad_clip_coef(g, n, 0.1, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"grad_clip_coef: {ms:.3f} ms = {n*4/ms/1e9:.2f} TB/s")
