"""What the epilogue variants of one forward GEMM cost (same launch, same tile): plain, + bias, + bias + GELU, + bias + GELU with the
pre-activation saved (training forward of fc1), + residual; python tools/bench_epilogue.py [M N K [tile hints ...]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
from bench_tile_ab import timeit

M, N, Kd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 4096, 1024)
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
x = r(M, Kd); w = r(N, Kd); bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
pre = torch.empty_like(out); res = r(M, N)
fl = 2.0 * M * N * Kd
hints = [int(v) for v in sys.argv[4:]] or [0]
for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+gelu", dict(bias=bias, act="gelu")), ("bias+quickgelu", dict(bias=bias, act="quick_gelu")),
                 ("bias+gelu+preact", dict(bias=bias, act="gelu", preact=pre)), ("bias+residual", dict(bias=bias, residual=res)),
                 ("dact (gelu')", dict(dact_src=res, act="gelu"))):
    line = f"{name:20s}"
    for h in hints:
        try:
            t = timeit(lambda: K.gemm(x, w, M, N, Kd, out=out, tile_hint=h, **kw))
            line += f"  [{h}] {t:8.1f} us {fl / t / 1e6:6.0f} TFLOP/s"
        except Exception as e:
            line += f"  [{h}] failed: {repr(e)[:120]}"
    print(line, flush=True)
