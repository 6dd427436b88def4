import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
from tools.bench_kernels import timeit
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
M, N = 16384, 1024
for Kd in (64, 1024):
    x = r(M, Kd); w = r(N, Kd); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for hint in (256, 3256, 6256, 8256, 8128, 8064):
        t = timeit(lambda: K.gemm(x, w, M, N, Kd, out=out, tile_hint=hint), iters=50)
        print(f"K={Kd:5d} t{hint}: {t*1e6:7.1f} us  {2*M*N*Kd/t/1e12:7.1f} TF/s")
# pure copy of the same output volume for reference
a = r(M, N)
t = timeit(lambda: K.add(a, a), iters=50)
print(f"add kernel over [16384,1024]: {t*1e6:.1f} us ({3*M*N*2/t/1e9:.0f} GB/s)")
