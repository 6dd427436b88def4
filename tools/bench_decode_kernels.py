"""Decode-step kernels one by one (run under tools/kt.sh with CHUNK=1): skinny GEMM shapes, single-query attention, top-k."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
sep = torch.zeros(1024, device=dev)
hint = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (M, N, Kd) in [(5, 1024, 1024), (5, 3072, 1024), (5, 4096, 1024), (5, 1024, 4096), (5, 50267, 1024)]:
    x = r(M, Kd); w = r(N, Kd); b = torch.randn(N, device=dev)
    out = torch.empty(M, N if N % 8 == 0 else 50272, device=dev, dtype=torch.float32 if N > 50000 else torch.bfloat16)
    sep.add_(1.0)
    for _ in range(20):
        K.gemm(x, w, M, N, Kd, bias=b, out=out, ldo=out.shape[1], out_mode=1 if N > 50000 else 0, tile_hint=hint)
    print("gemm", M, N, Kd)
for Tk in (25, 512):
    q = r(5, 1, 1024); kv = r(5, Tk, 2048); mask = torch.ones(5, Tk, dtype=torch.uint8, device=dev)
    sep.add_(1.0)
    for _ in range(20):
        K.attn_fwd(q, kv[..., :1024], kv[..., 1024:], 5, 16, 1, Tk, key_mask=mask, need_lse=False)
    print("attn", Tk)
logits = torch.randn(5, 50272, device=dev); bs = torch.zeros(5, device=dev); bans = torch.full((5, 50), -1, dtype=torch.int32, device=dev)
sep.add_(1.0)
for _ in range(20):
    K.beam_topk(logits, 50267, 10, beam_scores=bs, bans=bans, eos=2, suppress_eos=True)
sep.add_(1.0)
torch.cuda.synchronize()
