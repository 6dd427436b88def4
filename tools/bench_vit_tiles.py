"""ViT-L/14 GEMM shapes at batch 32 (M = 32 * 257 = 8224 rows): tile choice vs occupancy.  256x256 tiles give 33 x N/256 tiles
(N = 1024: 132 on 256 CUs); 8192 rows in 256x128 tiles are exactly 256.  python tools/bench_vit_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K


def t(fn, n=20):
    for _ in range(3):
        fn()
    best = 1e30
    for _ in range(5):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


for N, Kd, kw in ((1024, 1024, "res"), (1024, 4096, "res"), (3072, 1024, ""), (4096, 1024, "act")):
    M = 8224
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").bfloat16(); o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    extra = dict(residual=res) if kw == "res" else dict(act="quick_gelu") if kw == "act" else {}
    line = f"N={N} K={Kd} {kw:4s}: default {t(lambda: K.gemm(x, w, M, N, Kd, bias=b, out=o, **extra)):7.1f} us"
    for hint in (256, 264, 128):
        line += f" | hint {hint} on 8224 rows {t(lambda: K.gemm(x, w, M, N, Kd, bias=b, out=o, tile_hint=hint, **extra)):7.1f}"
    for hint in (256, 264, 128):
        def split():
            ex1 = dict(extra); ex2 = dict(extra)
            if "residual" in extra:
                ex1["residual"] = res[:8192]; ex2["residual"] = res[8192:]
            K.gemm(x[:8192], w, 8192, N, Kd, bias=b, out=o[:8192], tile_hint=hint, **ex1)
            K.gemm(x[8192:], w, 32, N, Kd, bias=b, out=o[8192:], tile_hint=64, **ex2)
        line += f" | 8192 rows hint {hint} + 32 rows {t(split):7.1f}"
    print(line, flush=True)
