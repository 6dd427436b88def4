"""What bf16 GEMM rate is attainable on this box?  The hand-written tile kernels (vacnic_gemm_bf16) beside the vendor library
(torch.matmul -> hipBLASLt / rocBLAS) on the configs[1] shapes, all three layouts of a Linear (forward X W^T, dgrad dY W,
wgrad dY^T X).  The library is a yardstick only: nothing on the product path calls it.  HIP events, 20 launches, best of 5."""
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
    r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    shapes = [(16384, 1024, 1024), (16384, 3072, 1024), (16384, 4096, 1024), (16384, 1024, 4096), (8192, 1024, 1024), (8192, 4096, 1024),
              (8192, 1024, 4096), (2048, 1024, 1024), (2048, 4096, 1024), (2048, 1024, 4096), (8192, 8192, 8192), (4112, 1024, 1024), (4112, 4096, 1024)]
    print(f"{'M':>6s} {'N':>5s} {'K':>5s} | {'fwd ours':>9s} {'lib':>8s} | {'dgrad ours':>10s} {'lib':>8s} | {'wgrad ours':>10s} {'lib':>8s}   (TFLOP/s; us)")
    for M, N, Kd in shapes:
        x = r(M, Kd); w = r(N, Kd); dy = r(M, N)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dx = torch.empty(M, Kd, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(N, Kd, device="cuda"); dws = [torch.zeros(N, Kd, device="cuda") for _ in range(4)]; dwl = torch.empty(N, Kd, device="cuda", dtype=torch.bfloat16)
        wt = w.t()
        fl = 2.0 * M * N * Kd
        tiles = ((N + 127) // 128) * ((Kd + 127) // 128)
        t = {}
        t["fo"] = timeit(lambda: K.gemm(x, w, M, N, Kd, out=out))
        t["fl"] = timeit(lambda: torch.matmul(x, wt, out=out))
        t["do"] = timeit(lambda: K.gemm(dy, w, M, Kd, N, out=dx, ldw=Kd, w_kstrided=True))
        t["dl"] = timeit(lambda: torch.matmul(dy, w, out=dx))
        if M >= 1024 and N * Kd <= 4096 * 1024:
            t["wo"] = timeit(lambda: K.wgrad_group([(dy, x, d, None) for d in dws])) / 4 if M <= 4096 else timeit(
                lambda: K.gemm(dy, x, N, Kd, M, out=dw, ldx=N, ldw=Kd, ldo=Kd, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=K.wgrad_split(M, tiles)))
        else:           # outputs beyond the grouped kernel's job size: the plain weight-gradient GEMM
            t["wo"] = timeit(lambda: K.gemm(dy, x, N, Kd, M, out=dw, ldx=N, ldw=Kd, ldo=Kd, x_kstrided=True, w_kstrided=True, out_mode=2,
                                            split_k=K.wgrad_split(M, tiles)))
        dyt = dy.t()
        t["wl"] = timeit(lambda: torch.matmul(dyt, x, out=dwl))
        f = lambda k: f"{fl / t[k] / 1e6:6.0f} ({t[k]:6.1f})"
        print(f"{M:6d} {N:5d} {Kd:5d} | {f('fo')} {f('fl')} | {f('do')} {f('dl')} | {f('wo')} {f('wl')}", flush=True)


if __name__ == "__main__":
    main()
