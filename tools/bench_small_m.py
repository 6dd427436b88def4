"""GPU box: the forward GEMMs of ONE caption's encoder / ViT pass (batch 1: M = 512 text tokens, 257 patches) — the unsplit launch the
cost model picks against K slices through the ordered fix-up.  HIP events, 20 launches, best of 5; weights stay cache-resident."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
from bench_fixup import timeit

r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
for M in (512, 257, 90):
    for N, Kd, act in ((1024, 1024, None), (3072, 1024, None), (4096, 1024, "gelu"), (1024, 4096, None), (2048, 1024, None)):
        x, w = r(M, Kd), r(N, Kd)
        bias = torch.randn(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(out=out, bias=bias, act=act)
        t0 = timeit(lambda: K.gemm(x, w, M, N, Kd, tile_hint=-1, **kw))
        row = [f"M {M:4d} N {N:5d} K {Kd:5d} {act or '-':4s} | cost model {t0:6.1f} us |"]
        best = (t0, "cost model")
        for hint in (64, 128):
            bm, bn = {64: (64, 128), 128: (128, 128)}[hint]
            for sp in (1, 2, 4, 8, 16):
                if Kd // sp < 128:
                    continue
                wgs = ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * sp
                if wgs > 1024:
                    continue
                t = timeit(lambda: K.gemm(x, w, M, N, Kd, split_k=sp, fixup=sp > 1, tile_hint=hint, **kw))
                row.append(f"{hint}/{sp}:{t:5.1f}")
                if t < best[0]:
                    best = (t, f"{hint}/{sp}")
        print(" ".join(row), "| best", best[1], f"{best[0]:.1f} us", flush=True)
