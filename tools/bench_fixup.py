"""Split-K through the ordered fix-up against the shipped choice, per shape: HIP events, 20 launches, best of 5.
    python tools/bench_fixup.py            -> table on stdout
Layouts: NN forward, NT dgrad (W K-strided), TT weight gradient (accumulate into fp32)."""
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
    cases = [("NN", 16384, 1024, 1024, 0), ("NN", 16384, 4096, 1024, 0), ("NN", 2048, 1024, 4096, 0), ("NN", 640, 1024, 4096, 0), ("NN", 2048, 1024, 1024, 0), ("NN", 2048, 4096, 1024, 0),
             ("NT", 2048, 1024, 4096, 0), ("NT", 2048, 1024, 3072, 0), ("NT", 640, 1024, 4096, 0), ("NT", 2688, 1024, 2048, 0), ("NT", 2048, 1024, 1024, 0),
             ("NT", 4112, 1024, 4096, 0), ("NN", 4112, 1024, 4096, 0), ("NN", 8224, 1024, 4096, 0),
             ("NT", 2048, 1024, 16384, 2),
             ("TT", 1024, 1024, 16384, 2), ("TT", 3072, 1024, 16384, 2), ("TT", 4096, 1024, 16384, 2), ("TT", 1024, 4096, 16384, 2),
             ("TT", 1024, 1024, 2048, 2), ("TT", 4096, 1024, 2048, 2), ("TT", 1024, 1024, 8224, 2)]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for kind, M, N, Kd, om in cases:
        if only and kind != only:
            continue
        xk, wk = kind[0] == "T", kind[1] == "T"
        x = r(Kd, M) if xk else r(M, Kd)
        w = r(Kd, N) if wk else r(N, Kd)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if om else torch.bfloat16)
        res = r(M, N) if not om else None
        fl = 2.0 * M * N * Kd
        kw = dict(out=out, x_kstrided=xk, w_kstrided=wk, out_mode=om, residual=res)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        base_split = (K.wgrad_split(Kd, tiles) if kind == "TT" else 8) if om == 2 else 1
        t0 = timeit(lambda: K.gemm(x, w, M, N, Kd, split_k=base_split, **kw))
        row = [f"{kind} {M:5d} {N:5d} {Kd:5d} om{om} | shipped (split {base_split}{' atomics' if base_split > 1 else ''}) {t0:6.1f} us {fl / t0 / 1e6:5.0f} TF |"]
        best = (t0, "shipped")
        for hint in (64, 128, 264, 256):
            if hint in (256, 264) and (M < 256 or N < 256):
                continue
            for sp in (2, 4, 8, 16):
                if Kd // sp < 256:
                    continue
                bm, bn = {64: (64, 128), 128: (128, 128), 264: (256, 128), 256: (256, 256)}[hint]
                wgs = ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * sp
                if wgs > 1024 or wgs < 96:
                    continue
                t = timeit(lambda: K.gemm(x, w, M, N, Kd, split_k=sp, fixup=True, tile_hint=hint, **kw))
                row.append(f"{hint}/{sp}:{t:5.1f}")
                if t < best[0]:
                    best = (t, f"{hint}/{sp}")
        print(" ".join(row), f"| best {best[1]} {best[0]:.1f} us {fl / best[0] / 1e6:.0f} TF ({t0 / best[0]:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
