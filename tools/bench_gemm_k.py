"""GEMM tile-config A/B under rocprofv3 (tools/kt.sh): python tools/bench_gemm_k.py <hints,comma> [layout nn|nt|tn|tt] [M N K...]
Each (K, hint) case is separated from the next by unrelated kernels so tools/kstat.py (runs mode) prints one line per case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
HINTS = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (256, 128)
lay = sys.argv[2] if len(sys.argv) > 2 else "nn"
M, N = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16384, 1024)
Ks = tuple(int(v) for v in sys.argv[5:]) or (1024, 4096)
for Kd in Ks:
    x = r(M, Kd); w = r(N, Kd); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    xa = x.t().contiguous() if lay[0] == "t" else x
    wa = w.t().contiguous() if lay[1] == "t" else w
    kw = dict(x_kstrided=lay[0] == "t", w_kstrided=lay[1] == "t")
    if lay[0] == "t": kw["ldx"] = xa.shape[1]
    ref = x.float() @ w.float().t()
    for hint in HINTS:
        out.zero_(); K.gemm(xa, wa, M, N, Kd, out=out, tile_hint=hint, **kw); torch.cuda.synchronize()
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        print(f"{lay} K={Kd} hint {hint}: rel err {err:.2e}")
        for _ in range(40): K.gemm(xa, wa, M, N, Kd, out=out, tile_hint=hint, **kw)
        torch.cuda.synchronize()
