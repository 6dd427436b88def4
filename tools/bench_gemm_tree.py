"""Same-box GEMM comparison ACROSS source trees: VACNIC_TREE=<dir> python tools/bench_gemm_tree.py   (auto tile choice, the
shapes of one training step).  Prints one line per shape: median us over 5 rounds of 30 launches."""
import os
import statistics as st
import sys
tree = os.environ.get("VACNIC_TREE", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.abspath(tree))
import torch
from vacnic_amd import kernels as K

dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()


def timed(fn, iters=30):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


CASES = [("NN", 16384, 1024, 1024, {}), ("NN", 16384, 1024, 1024, {"residual": 1}), ("NN", 16384, 3072, 1024, {}), ("NN", 16384, 4096, 1024, {"act": "gelu"}),
         ("NN", 16384, 4096, 1024, {"act": "gelu", "preact": 1}), ("NN", 16384, 1024, 4096, {}), ("NN", 8224, 4096, 1024, {"act": "quick_gelu"}),
         ("NN", 8224, 1024, 4096, {"residual": 1}), ("NN", 8224, 3072, 1024, {}), ("NN", 2048, 1024, 1024, {}), ("NN", 2048, 4096, 1024, {"act": "gelu", "preact": 1}),
         ("NT", 16384, 1024, 1024, {}), ("NT", 16384, 1024, 1024, {"residual": 1}), ("NT", 16384, 4096, 1024, {"dact": 1}), ("NT", 16384, 1024, 4096, {}),
         ("NT", 16384, 1024, 4096, {"residual": 1}), ("NT", 16384, 1024, 3072, {}), ("NT", 16384, 1024, 3072, {"residual": 1}), ("NT", 2048, 1024, 1024, {}),
         ("TT", 1024, 1024, 16384, {}), ("TT", 3072, 1024, 16384, {}), ("TT", 1024, 4096, 16384, {}), ("TT", 4096, 1024, 16384, {}), ("TT", 1024, 1024, 2048, {})]
tot = 0.0
for lay, M, N, Kd, epi in CASES:
    x = r(M, Kd); w = r(N, Kd)
    kw = {}
    if lay == "NT":
        w = w.t().contiguous(); kw["w_kstrided"] = True
    if lay == "TT":
        x = x.t().contiguous(); w = w.t().contiguous(); kw.update(x_kstrided=True, w_kstrided=True, out_mode=2)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        kw["split_k"] = K.wgrad_split(Kd, tiles)
        out = torch.zeros(M, N, device=dev)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw["bias"] = torch.randn(N, device=dev) if lay == "NN" else None
    if epi.get("residual"):
        kw["residual"] = r(M, N)
    if epi.get("act"):
        kw["act"] = epi["act"]
    if epi.get("preact"):
        kw["preact"] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if epi.get("dact"):
        kw["dact_src"] = r(M, N); kw["act"] = "gelu"
    f = lambda: K.gemm(x, w, M, N, Kd, out=out, **kw)
    f(); torch.cuda.synchronize()
    t = st.median(timed(f) for _ in range(5))
    tot += t
    print(f"{lay} {M:6d}x{N:5d}x{Kd:6d} {str(sorted(epi)):22s} {t:8.1f} us {2.0 * M * N * Kd / t / 1e6:7.0f} TF", flush=True)
print(f"sum {tot:.1f} us")
