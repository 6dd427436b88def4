"""Tile configurations side by side on the configs[1] GEMM shapes: python tools/bench_tile_ab.py 256 264 [...]
forward (X W^T + bias, GELU), dgrad (dY W, W K-strided) and weight gradient (both K-strided, fp32 accumulate, split-K 4) layouts;
prints us / TFLOP/s per hint and the largest difference of each hint's result from the first hint's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K


def timeit(fn, n=20):
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


def main():
    hints = [int(v) for v in sys.argv[1:]] or [256, 264]
    r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    shapes = [(16384, 1024, 1024), (16384, 3072, 1024), (16384, 4096, 1024), (16384, 1024, 4096), (8192, 4096, 1024), (4100, 1024, 1024), (8192, 8192, 8192)]
    for M, N, Kd in shapes:
        x = r(M, Kd); w = r(N, Kd); dy = r(M, N); bias = torch.randn(N, device="cuda")
        fl = 2.0 * M * N * Kd
        line = f"{M:6d} {N:5d} {Kd:5d} |"
        ref = {}
        for h in hints:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dx = torch.empty(M, Kd, device="cuda", dtype=torch.bfloat16)
            dw = torch.zeros(N, Kd, device="cuda"); db = torch.zeros(N, device="cuda")
            f = lambda: K.gemm(x, w, M, N, Kd, out=out, bias=bias, act="gelu", tile_hint=h)
            d = lambda: K.gemm(dy, w, M, Kd, N, out=dx, ldw=Kd, w_kstrided=True, tile_hint=h)
            g = lambda: K.gemm(dy, x, N, Kd, M, out=dw, ldx=N, ldw=Kd, ldo=Kd, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=4, xsum=db, tile_hint=h)
            tf, td = timeit(f), timeit(d)
            tg = timeit(g) if N * Kd <= 4096 * 1024 else float("nan")
            dw.zero_(); db.zero_(); f(); d()
            if N * Kd <= 4096 * 1024:
                g()
            torch.cuda.synchronize()
            res = (out.float(), dx.float(), dw.clone(), db.clone())
            # run-to-run: the bf16 results of 10 more launches must be bit-identical (a race in a ring / barrier scheme shows here)
            for _ in range(10):
                f(); d()
                torch.cuda.synchronize()
                if not (torch.equal(out.float(), res[0]) and torch.equal(dx.float(), res[1])):
                    line += f"  [{h}] NOT REPRODUCIBLE"
                    break
            if not ref:
                ref = res
            err = [((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(res, ref)]
            line += f"  [{h}] fwd {tf:7.1f} us {fl / tf / 1e6:5.0f} | dgrad {td:7.1f} {fl / td / 1e6:5.0f} | wgrad {tg:7.1f} {fl / tg / 1e6:5.0f} | err {max(err):.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
