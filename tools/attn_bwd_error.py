"""How accurate is the attention backward when attention is nearly uniform (random-init model)?  dQ / dK are then covariances
over the keys — sums of zero-mean terms — and the bf16 rounding of dS before the dS.K / dS^T.Q products does not cancel with them.
python tools/attn_bwd_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K


def run(B, H, Tq, Tk, qk_scale, causal, cm=0.0):
    """cm: size of a component shared by all value rows / all output-gradient rows (hidden states of one sequence are mostly parallel)"""
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda().bfloat16()
    q, k = mk(B, Tq, H * 64, sc=qk_scale), mk(B, Tk, H * 64, sc=qk_scale)
    v = (torch.randn(B, Tk, H * 64, generator=g) * 0.5 + cm * torch.randn(1, 1, H * 64, generator=g)).cuda().bfloat16()
    dout = (torch.randn(B, Tq, H * 64, generator=g) + cm * torch.randn(1, 1, H * 64, generator=g)).cuda().bfloat16()
    out, lse = K.attn_fwd(q, k, v, B, H, Tq, Tk, causal=causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, causal=causal)
    qa, ka, va = (t.float().view(B, -1, H, 64).transpose(1, 2).clone().requires_grad_(True) for t in (q, k, v))
    s = qa @ ka.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf"), device="cuda"), 1)
    o = torch.softmax(s, -1) @ va
    o.backward(dout.float().view(B, Tq, H, 64).transpose(1, 2))
    res = []
    for got, ref in ((dq, qa.grad), (dk, ka.grad), (dv, va.grad)):
        gg = got.float().view(B, -1, H, 64).transpose(1, 2)
        res.append(((gg - ref).norm() / ref.norm()).item())
    return res


for name, Tq, Tk, causal in (("decoder self (64 x 64, causal)", 64, 64, True), ("decoder cross (64 x 512)", 64, 512, False),
                             ("encoder self (512 x 512)", 512, 512, False)):
    for sc, cm in ((1.0, 0.0), (0.2, 0.0), (0.2, 1.0), (0.2, 3.0)):
        e = run(4, 16, Tq, Tk, sc, causal, cm)
        print(f"[VACNIC_ATTN_DELTA={os.environ.get('VACNIC_ATTN_DELTA', '0')}] {name:32s} q,k std {sc:4.2f} common component {cm:3.1f}: "
              f"rel err dq {e[0]:.4f} dk {e[1]:.4f} dv {e[2]:.4f}", flush=True)
