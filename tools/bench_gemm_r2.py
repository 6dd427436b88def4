"""Round-2 GEMM A/B on the 256-row ping-pong tiles: persistent launch + bf16 single-pass epilogue vs the round-1 forms.
   python tools/bench_gemm_r2.py [quick]
Variants (tile_hint = 256 + 1000 * debug bits): new = persistent + bf16 epilogue; np = one workgroup per tile (bit 6);
old = fp32 four-pass epilogue (bit 4), one workgroup per tile = the round-1 kernel.  HIP events on the launch stream,
interleaved rounds in one process (median over rounds)."""
import os
import statistics as st
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K

dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
VARIANTS = (("new", 256), ("np", 256 + 64000), ("old-epi", 256 + 64000 + 16000), ("half-line DMA (wrong results)", 256 + 512000))
CHECK = {"new", "np", "old-epi"}


def timed(fn, iters):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def case(name, M, N, Kd, lay="nn", **epi):
    x = r(M, Kd); w = r(N, Kd)
    xa = x.t().contiguous() if lay[0] == "t" else x
    wa = w.t().contiguous() if lay[1] == "t" else w
    kw = dict(x_kstrided=lay[0] == "t", w_kstrided=lay[1] == "t")
    if lay[0] == "t":
        kw["ldx"] = xa.shape[1]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if epi.get("bias"):
        kw["bias"] = torch.randn(N, device=dev)
    if epi.get("residual"):
        kw["residual"] = r(M, N)
    if epi.get("act"):
        kw["act"] = epi["act"]
    if epi.get("preact"):
        kw["preact"] = torch.empty_like(out)
    if epi.get("dact"):
        kw["dact_src"] = r(M, N); kw["act"] = "gelu"
    ref = None
    res = {}
    for vn, hint in VARIANTS:
        out.zero_()
        K.gemm(xa, wa, M, N, Kd, out=out, tile_hint=hint, **kw)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.float().clone()
        elif vn in CHECK:
            d = (out.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            assert d < 2e-2, f"{name} {vn}: variants disagree by {d}"
    rounds = 5
    for _ in range(rounds):
        for vn, hint in VARIANTS:
            res.setdefault(vn, []).append(timed(lambda: K.gemm(xa, wa, M, N, Kd, out=out, tile_hint=hint, **kw), 30))
    fl = 2.0 * M * N * Kd
    line = f"{name:44s}"
    for vn, _ in VARIANTS:
        t = st.median(res[vn])
        line += f" | {vn} {t:7.1f} us {fl / t / 1e6:7.0f} TF"
    print(line, flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    case("xattn kv/q/out  NN 16384x1024x1024 bias", 16384, 1024, 1024, bias=True)
    case("out_proj        NN 16384x1024x1024 bias+res", 16384, 1024, 1024, bias=True, residual=True)
    case("dec cross k|v   NN 16384x2048x1024 bias", 16384, 2048, 1024, bias=True)
    case("self k|v|q      NN 16384x3072x1024 bias", 16384, 3072, 1024, bias=True)
    case("fc1 guide       NN 16384x4096x1024 gelu", 16384, 4096, 1024, bias=True, act="gelu")
    case("fc1 student     NN 16384x4096x1024 gelu+preact", 16384, 4096, 1024, bias=True, act="gelu", preact=True)
    case("fc2             NN 16384x1024x4096 bias+res", 16384, 1024, 4096, bias=True, residual=True)
    if quick:
        return
    case("batched cross   NN 16384x24576x1024 bias", 16384, 24576, 1024, bias=True)
    case("vit fc1         NN 8192x4096x1024 qgelu", 8192, 4096, 1024, bias=True, act="quick_gelu")
    case("vit qkv         NN 8192x3072x1024 bias", 8192, 3072, 1024, bias=True)
    case("dgrad fc2       NT 16384x4096x1024 dact", 16384, 4096, 1024, lay="nt", dact=True)
    case("dgrad fc1       NT 16384x1024x4096", 16384, 1024, 4096, lay="nt")
    case("dgrad out       NT 16384x1024x1024", 16384, 1024, 1024, lay="nt")
    case("dgrad kvq       NT 16384x1024x3072", 16384, 1024, 3072, lay="nt")
    case("batched dgrad   NT 16384x1024x24576", 16384, 1024, 24576, lay="nt")


if __name__ == "__main__":
    main()
