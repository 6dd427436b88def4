"""config 5 (batch 1, beam 5, 50 tokens): where does a caption's wall time go?  Synchronised timers around the stages of generate()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import generate as Gn, kernels as K, synthetic
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.models.clip_vit import extract_clip_img_feat, graphed_clip_img_feat
from vacnic_amd.training import build_models, to_device

cfg, vcfg = bart_large_vit_l14()
model, _, clip_model = build_models(cfg, vcfg, device="cuda", seed=42, init="device", with_guide=False)
model.eval()
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append(time.perf_counter() - t0); return time.perf_counter()
orig_enc = Gn.GraphedCall.__call__
orig_begin = Gn.CachedDecoder.begin
orig_run = Gn.DecodeSession.run_device
def enc(self, *a):
    if len(a) < 5:                      # the ViT's graphed call
        return orig_enc(self, *a)
    t0 = tick("pre-encoder (host)", T["_t"]); r = orig_enc(self, *a); T["_t"] = tick("encoder", t0); return r
def begin(self, *a, **kw):
    t0 = tick("session lookup", T["_t"]); r = orig_begin(self, *a, **kw); T["_t"] = tick("begin (cross K/V)", t0); return r
def run(self, *a, **kw):
    t0 = tick("beam init", T["_t"]); r = orig_run(self, *a, **kw); T["_t"] = tick("decode loop + readback", t0); return r
Gn.GraphedCall.__call__ = enc; Gn.CachedDecoder.begin = begin; Gn.DecodeSession.run_device = run
with torch.no_grad():
    for i in range(8):
        b = to_device(synthetic.make_batch(cfg, 1, S=512, T=64, seed=42, step=i, full_length=True), "cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter(); ta = t0
        mask, _ = K.prep_ids(b["article_ids"], 1)
        nmask, _ = K.prep_ids(b["names_art_ids"], 1)
        t0 = tick("prep", t0)
        _, cls = graphed_clip_img_feat(clip_model)(b["img_tensor"])
        T["_t"] = tick("ViT", t0)
        out = model.generate(input_ids=b["article_ids"], attention_mask=mask, num_beams=5, max_length=50, length_penalty=2.0, min_length=49,
                             image_features=cls, face_features=b["face_emb"], face_mask=K.face_mask(b["face_emb"]),
                             name_ids=b["names_art_ids"], name_mask=nmask, add_ner_ffn=True)
        tick("finalize (host)", T["_t"])
        T.setdefault("total", []).append(time.perf_counter() - ta)
for k, v in T.items():
    if k != "_t":
        print(f"{k:28s} {sum(v[3:]) / len(v[3:]) * 1e3:8.2f} ms")
ses = next(iter(model._decode_sessions.values()))
print("S (encoder length) =", ses.dec.S)
