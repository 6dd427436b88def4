"""LM-head + CE pieces at configs[1] size (R = 2048, V = 50267, d = 1024): fused forward, one dlogits chunk, dh / dE chunk GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
from tools.bench_kernels import timeit
dev = "cuda"
R, V, d, CH = 2048, 50267, 1024, 16384
Vp = (V + 31) // 32 * 32
h = (torch.randn(R, d, device=dev) * 0.5).bfloat16()
E = torch.zeros(Vp, d, device=dev, dtype=torch.bfloat16); E[:V] = (torch.randn(V, d, device=dev) * 0.05).bfloat16()
tgt = torch.randint(3, V, (R,), device=dev)
lse, acc = K.lmhead_ce_fwd(h, E, tgt, V)
rowp = K.lmhead_ce_rowp(lse, tgt, acc)
dl = torch.empty(R, CH, device=dev, dtype=torch.bfloat16)
dh32 = torch.zeros(R, d, device=dev)
eg = torch.zeros(Vp, d, device=dev)
logits = torch.empty(R, Vp, device=dev, dtype=torch.float32)
print("plain logits GEMM f32  %7.1f us" % (timeit(lambda: K.gemm(h, E, R, V, d, out=logits, ldo=Vp, out_mode=1)) * 1e6))
print("fused fwd (stats)      %7.1f us" % (timeit(lambda: K.lmhead_ce_fwd(h, E, tgt, V)) * 1e6))
print("dlogits chunk 16384    %7.1f us" % (timeit(lambda: K.lmhead_ce_dlogits(h, E, tgt, V, rowp, dl, 0, CH)) * 1e6))
print("plain bf16 GEMM chunk  %7.1f us" % (timeit(lambda: K.gemm(h, E[:CH], R, CH, d, out=dl, ldo=CH)) * 1e6))
print("dh chunk (NT splitK 8) %7.1f us" % (timeit(lambda: K.gemm(dl, E[:CH], R, d, CH, out=dh32, ldx=CH, w_kstrided=True, out_mode=2, split_k=8)) * 1e6))
tiles = ((CH + 127) // 128) * ((d + 127) // 128)
print("dE chunk (TT)          %7.1f us" % (timeit(lambda: K.gemm(dl, h, CH, d, R, out=eg[:CH], ldx=CH, ldw=d, ldo=d, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=K.wgrad_split(R, tiles))) * 1e6))
