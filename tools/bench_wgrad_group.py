"""Grouped weight gradients (vacnic_wgrad_group: one XCD per 1024 x 1024 output block, full reduction, no atomics) against one
split-K GEMM per Linear (fp32 atomics), on the weight-gradient shapes of configs[1].  HIP events on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K


def timeit(fn, n=10):
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
    cases = [("encoder d x d (out / q / cross-out)", 16384, 1024, 1024, 8), ("encoder k|v|q", 16384, 3072, 1024, 4), ("encoder fc1", 16384, 4096, 1024, 4),
             ("encoder fc2", 16384, 1024, 4096, 4), ("decoder d x d", 2048, 1024, 1024, 16), ("decoder k|v|q", 2048, 3072, 1024, 4),
             ("decoder fc1", 2048, 4096, 1024, 4), ("decoder fc2", 2048, 1024, 4096, 4)]
    for name, M, N, Kd, njobs in cases:
        jobs = []
        for j in range(njobs):
            dy = (torch.randn(M, N, device="cuda") * 0.1).bfloat16(); x = torch.randn(M, Kd, device="cuda").bfloat16()
            jobs.append((dy, x, torch.zeros(N, Kd, device="cuda"), torch.zeros(N, device="cuda")))
        tiles = ((N + 127) // 128) * ((Kd + 127) // 128)

        def old():
            for dy, x, dw, db in jobs:
                K.gemm(dy, x, N, Kd, M, out=dw, ldx=N, ldw=Kd, ldo=Kd, x_kstrided=True, w_kstrided=True, out_mode=2,
                       split_k=K.wgrad_split(M, tiles), xsum=db)
        t_old = timeit(old)
        t_new = timeit(lambda: K.wgrad_group(jobs))
        fl = 2.0 * M * N * Kd * njobs
        print(f"{name:40s} M={M:6d} N={N:5d} K={Kd:5d} x{njobs:2d}: split-K {t_old / njobs:8.1f} us/job {fl / t_old / 1e6:7.1f} TF/s | grouped {t_new / njobs:8.1f} us/job "
              f"{fl / t_new / 1e6:7.1f} TF/s  ({t_old / t_new:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
