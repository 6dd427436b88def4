"""Micro-benchmarks of the hot kernels at BASELINE cfg2 shapes (HIP events on the launch stream)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    res = []
    dev = "cuda"
    def r(*s):
        return (torch.randn(*s, device=dev) * 0.5).bfloat16()
    # GEMMs
    for name, M, N, Kd in [("xattn_kv 16384x1024x1024", 16384, 1024, 1024), ("qkv 16384x3072x1024", 16384, 3072, 1024),
                           ("fc1 16384x4096x1024", 16384, 4096, 1024), ("fc2 16384x1024x4096", 16384, 1024, 4096),
                           ("dec 2048x1024x1024", 2048, 1024, 1024), ("lmhead 2048x50267x1024", 2048, 50267, 1024),
                           ("vit 8224x4096x1024", 8224, 4096, 1024), ("face 128x1024x3072", 128, 1024, 3072),
                           ("img 640x1024x4096", 640, 1024, 4096), ("names 2560x1024x1024", 2560, 1024, 1024)]:
        x = r(M, Kd); w = r(N, Kd); b = torch.zeros(N, device=dev)
        ldo = (N + 31) // 32 * 32
        out = torch.empty(M, ldo, device=dev, dtype=torch.bfloat16)
        for hint in (256, 260, 128, 261):
            t = timeit(lambda: K.gemm(x, w, M, N, Kd, bias=b, out=out, ldo=ldo, tile_hint=hint))
            res.append((f"gemm fwd t{hint} {name}", t, 2 * M * N * Kd / t / 1e12, "TF/s"))
        if N <= 4096:
            dy = r(M, N)
            dx = torch.empty(M, Kd, device=dev, dtype=torch.bfloat16)
            for hint in (256, 260, 128, 261):
                t = timeit(lambda: K.gemm(dy, w, M, Kd, N, out=dx, w_kstrided=True, tile_hint=hint))
                res.append((f"gemm dgrad t{hint} {name}", t, 2 * M * N * Kd / t / 1e12, "TF/s"))
            dw = torch.zeros(N, Kd, device=dev)
            for hint, sp in ((128, K.wgrad_split(M, ((N + 127) // 128) * ((Kd + 127) // 128))), (261, K.wgrad_split(M, ((N + 127) // 128) * ((Kd + 127) // 128))), (256, max(1, 256 // (((N + 255) // 256) * ((Kd + 255) // 256)))), (260, max(1, 256 // (((N + 255) // 256) * ((Kd + 255) // 256))))):
                t = timeit(lambda: K.gemm(dy, x, N, Kd, M, out=dw, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=sp, tile_hint=hint))
                res.append((f"gemm wgrad t{hint} split {sp} {name}", t, 2 * M * N * Kd / t / 1e12, "TF/s"))
    # attention
    for name, B, H, Tq, Tk, causal in [("enc self 512x512", 32, 16, 512, 512, False), ("enc cross 512x40", 32, 16, 512, 40, False),
                                       ("dec self 64x64", 32, 16, 64, 64, True), ("dec cross 64x512", 32, 16, 64, 512, False),
                                       ("vit 257x257", 32, 16, 257, 257, False)]:
        d = H * 64
        q = r(B, Tq, d); kv = r(B, Tk, 2 * d); k = kv[..., :d]; v = kv[..., d:]
        mask = torch.ones(B, Tk, device=dev, dtype=torch.uint8)
        t = timeit(lambda: K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=mask, causal=causal))
        fl = 4 * B * H * Tq * Tk * 64
        res.append((f"attn fwd {name}", t, fl / t / 1e12, "TF/s"))
        out, lse = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=mask, causal=causal)
        do = r(B, Tq, d); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        t = timeit(lambda: K.attn_bwd(q, k, v, out, do, lse, dq, dkv[..., :d], dkv[..., d:], B, H, Tq, Tk, key_mask=mask, causal=causal))
        res.append((f"attn bwd {name}", t, 2.5 * fl / t / 1e12, "TF/s(5 products)"))
    # HBM-bound
    R, D = 16384, 1024
    x = r(R, D); rs = r(R, D); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    t = timeit(lambda: K.add_ln_fwd(x, rs, g, b))
    res.append(("add_ln fwd 16384x1024", t, 3 * R * D * 2 / t / 1e9, "GB/s"))
    t = timeit(lambda: K.add_ln_fwd(x, rs, g, b, p_drop=0.1, seed=1))
    res.append(("add_ln fwd +dropout", t, 3 * R * D * 2 / t / 1e9, "GB/s"))
    out, mean, rstd = K.add_ln_fwd(x, rs, g, b)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    t = timeit(lambda: K.add_ln_bwd(x, x, rs, g, mean, rstd, dg, db))
    res.append(("add_ln bwd 16384x1024", t, 4 * R * D * 2 / t / 1e9, "GB/s"))
    n = 256 * 1024 * 1024
    p = torch.zeros(n, device=dev); gg = torch.zeros(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    p16 = torch.empty(n, device=dev, dtype=torch.bfloat16); hy = torch.tensor([1e-4, 1.0], device=dev)
    t = timeit(lambda: K.adamw(p, gg, m, v, p16, hy, n), iters=5)
    res.append(("adamw 256M params", t, n * 34 / t / 1e9, "GB/s"))
    logits = r(2048, 50272); tgt = torch.randint(0, 50267, (2048,), device=dev)
    t = timeit(lambda: K.ce_fwd(logits, tgt, 50267))
    res.append(("ce fwd 2048x50267", t, 2048 * 50267 * 2 / t / 1e9, "GB/s"))
    for name, t, v, u in res:
        print(f"{name:48s} {t*1e6:10.1f} us  {v:9.1f} {u}")


if __name__ == "__main__":
    main()
