"""Attention forward / backward at the step's shapes, with and without attention dropout: HIP events, 20 launches, best of 5.
    python tools/bench_attn.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K


def timeit(fn, n=20, rounds=5):
    for _ in range(3):
        fn()
    best = 1e30
    for _ in range(rounds):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def main():
    r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    cnt = torch.zeros(1, device="cuda", dtype=torch.int64)
    shapes = [("enc self", 32, 16, 512, 512, False, True), ("enc cross img", 32, 16, 512, 40, False, False), ("dec self", 32, 16, 64, 64, True, False),
              ("dec cross", 32, 16, 64, 512, False, True), ("ViT", 32, 16, 257, 257, False, False), ("cfg4 self", 32, 16, 1024, 1024, False, True)]
    print(f"{'shape':14s} {'B':>3s} {'H':>3s} {'Tq':>5s} {'Tk':>5s} | fwd us (TF) | fwd+drop | bwd us (TF) | bwd+drop || every tile with one masked key (the general softmax path): fwd | bwd")
    for name, B, H, Tq, Tk, causal, masked in shapes:
        d = H * 64
        q = r(B, Tq, d); kv = r(B, Tk, 2 * d); do = r(B, Tq, d)
        km = torch.ones(B, Tk, device="cuda", dtype=torch.uint8) if masked else None
        fl = 4.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)
        res = []
        for p in (0.0, 0.1):
            kw = dict(key_mask=km, causal=causal, scale=0.125, p_drop=p, seed=7, seed_dev=cnt if p > 0 else None)
            out, lse = K.attn_fwd(q, kv[..., :d], kv[..., d:], B, H, Tq, Tk, need_lse=True, **kw)
            tf = timeit(lambda: K.attn_fwd(q, kv[..., :d], kv[..., d:], B, H, Tq, Tk, need_lse=True, **kw))
            dq = torch.empty_like(q); dkv = torch.empty_like(kv)
            tb = timeit(lambda: K.attn_bwd(q, kv[..., :d], kv[..., d:], out, do, lse, dq, dkv[..., :d], dkv[..., d:], B, H, Tq, Tk, **kw))
            res.append((tf, tb))
        (f0, b0), (f1, b1) = res
        km2 = torch.ones(B, Tk, device="cuda", dtype=torch.uint8); km2[:, 31::32] = 0       # a masked key in every 32-key block
        kw = dict(key_mask=km2, causal=causal, scale=0.125)
        out, lse = K.attn_fwd(q, kv[..., :d], kv[..., d:], B, H, Tq, Tk, need_lse=True, **kw)
        f2 = timeit(lambda: K.attn_fwd(q, kv[..., :d], kv[..., d:], B, H, Tq, Tk, need_lse=True, **kw))
        b2 = timeit(lambda: K.attn_bwd(q, kv[..., :d], kv[..., d:], out, do, lse, dq, dkv[..., :d], dkv[..., d:], B, H, Tq, Tk, **kw))
        print(f"{name:14s} {B:3d} {H:3d} {Tq:5d} {Tk:5d} | {f0:6.1f} ({fl / f0 / 1e6:5.0f}) | {f1:6.1f} ({fl / f1 / 1e6:5.0f}) | {b0:6.1f} ({2.5 * fl / b0 / 1e6:5.0f}) | {b1:6.1f} ({2.5 * fl / b1 / 1e6:5.0f}) || {f2:6.1f} | {b2:6.1f}", flush=True)


if __name__ == "__main__":
    main()
