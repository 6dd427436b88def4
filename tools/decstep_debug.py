"""debug aid: step kernel vs kernel-per-op chain, which rows / buffers differ and is the step kernel deterministic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import generate as Gn, kernels as K, synthetic
from vacnic_amd.config import ClipVisionConfig, VacnicConfig
from vacnic_amd.training import build_models

base = dict(d_model=768, encoder_layers=1, decoder_layers=2, encoder_attention_heads=12, decoder_attention_heads=12,
            encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=768, dropout=0.0)
cfg = VacnicConfig(**base)
R, nb, S, Tmax = 6, 3, 24, 12
vcfg = ClipVisionConfig(width=128, layers=1, patch_size=16, image_size=32, output_dim=64)
sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
sd["model.shared.weight"] = sd["model.shared.weight"] * synthetic.GEN_SHARPEN
model, _, _ = build_models(cfg, vcfg, init="synthetic", state_dicts=(sd, synthetic.make_state_dict(synthetic.guide_bart_param_shapes(cfg), seed=2),
                                                                      synthetic.make_state_dict(synthetic.clip_visual_param_shapes(vcfg), seed=4, std=0.05)))
model.eval()
B = R // nb
g = torch.Generator().manual_seed(5)
enc_h = (torch.randn(B, S, cfg.d_model, generator=g) * 0.7).bfloat16().cuda()
mask = torch.ones(B, S, dtype=torch.uint8); mask[:, S - 5:] = 0; mask = mask.cuda()
fast = Gn.CachedDecoder(model, R, S, Tmax, reorders=True)
fast2 = Gn.CachedDecoder(model, R, S, Tmax, reorders=True)
os.environ["VACNIC_DECODE_PER_OP"] = "1"
ref = Gn.CachedDecoder(model, R, S, Tmax, reorders=True)
cap = {}
orig = K.gemv_ln
orig_attn = K.attn_fwd
def spy(x, residual, *a, **kw):
    if kw.get("out_mode") == 1:
        cap["o"], cap["h"] = x.clone(), residual.clone()
    y = orig(x, residual, *a, **kw)
    if kw.get("act") == "gelu":
        cap["f"] = y.clone()
    elif kw.get("out_mode", 0) == 0 and kw.get("out") is None:
        cap["q"] = y.clone()          # last one = last layer's cross-attention query
        cap["o_so"], cap["h_in"] = x.clone(), residual.clone()
    return y
def spy_attn(q, *a, **kw):
    out = orig_attn(q, *a, **kw)
    cap.setdefault("ctx", []).append(out[0].clone())
    return out
with torch.no_grad():
    for dcd in (fast, fast2, ref):
        dcd.begin(enc_h, mask, nb)
    for t in range(Tmax - 1):
        ids = torch.randint(3, cfg.vocab_size, (R, 1), generator=g).cuda()
        if t > 0:
            src = torch.randint(0, nb, (R,), generator=g)
            src = (src + (torch.arange(R) // nb) * nb).cuda()
            for dcd in (fast, fast2, ref):
                dcd.reorder(src, t)
        la = fast.step(ids, t); fo, fh = fast.obuf.clone(), fast.hbuf[fast.L & 1].clone()
        la2 = fast2.step(ids, t); f2o = fast2.obuf.clone()
        K.gemv_ln = spy; K.attn_fwd = spy_attn
        cap["ctx"] = []
        lb = ref.step(ids, t)
        K.gemv_ln = orig; K.attn_fwd = orig_attn
        torch.cuda.synchronize()
        do = (fo.float() - cap["o"].float()).abs(); dh = (fh.float() - cap["h"].float()).abs()
        def nz(a, b):
            dd = (a.float().reshape(R, -1) - b.float().reshape(R, -1)).abs()
            return f"{dd.max().item():.2e}@{torch.nonzero(dd > 0)[:3].tolist()}"
        print(f"   cross ctx {nz(fast.ctxb, cap['ctx'][-1])}  q {nz(fast.qbuf, cap['q'])}  f {nz(fast.fbuf, cap['f'])}")
        print(f"t={t}: logits equal {torch.equal(la[:, :model.V], lb[:, :model.V])}  fast==fast2 {torch.equal(la, la2)} o==o2 {torch.equal(fo, f2o)}  "
              f"o diff max {do.max().item():.3e} at {torch.nonzero(do > 0)[:6].tolist()}  h diff max {dh.max().item():.3e} at {torch.nonzero(dh > 0)[:6].tolist()}  "
              f"cache eq {torch.equal(fast.cache_at(t)[:, :, :t + 1], ref.cache_at(t)[:, :, :t + 1])}")
