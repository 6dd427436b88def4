import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
from tools.bench_kernels import timeit
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
M, N, Kd = 16384, 4096, 1024
x = r(M, Kd); w = r(N, Kd); b = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); pre = torch.empty_like(out)
for hint in (128, 256):
    for name, kw in (("plain", {}), ("gelu", dict(act="gelu")), ("gelu+preact", dict(act="gelu", preact=pre)), ("quick_gelu", dict(act="quick_gelu"))):
        t = timeit(lambda: K.gemm(x, w, M, N, Kd, bias=b, out=out, tile_hint=hint, **kw))
        print(f"fc1 fwd t{hint} {name:12s} {t*1e6:8.1f} us {2*M*N*Kd/t/1e12:7.1f} TF/s")
dy = r(M, 1024); w2 = r(1024, N); du = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for hint in (128, 256):
    for name, kw in (("plain", {}), ("dact gelu", dict(act="gelu", dact_src=pre))):
        t = timeit(lambda: K.gemm(dy, w2, M, N, 1024, out=du, w_kstrided=True, tile_hint=hint, **kw))
        print(f"fc2 dgrad t{hint} {name:12s} {t*1e6:8.1f} us {2*M*N*1024/t/1e12:7.1f} TF/s")
res = r(M, 1024); o2 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16); h = r(M, N); w3 = r(1024, N)
for hint in (128, 256):
    for name, kw in (("plain", {}), ("residual", dict(residual=res))):
        t = timeit(lambda: K.gemm(h, w3, M, 1024, N, bias=torch.zeros(1024, device=dev), out=o2, tile_hint=hint, **kw))
        print(f"fc2 fwd t{hint} {name:12s} {t*1e6:8.1f} us {2*M*N*1024/t/1e12:7.1f} TF/s")
