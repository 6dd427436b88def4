"""GPU box: config 5 (batch 1, beam 5, 50 tokens) over a stream of captions — the sequential loop beside generate.CaptionPipeline
(caption i + 1's image tower, encoder and cross K/V on a side stream during caption i's beam search)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K, synthetic
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.generate import CaptionPipeline
from vacnic_amd.models.clip_vit import graphed_clip_img_feat
from vacnic_amd.training import build_models, to_device

cfg, vcfg = bart_large_vit_l14()
model, _, clip_model = build_models(cfg, vcfg, device="cuda", seed=42, init="device", with_guide=False)
model.eval()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
batches = [to_device(synthetic.make_batch(cfg, 1, S=512, T=64, seed=42, step=i, full_length=True), "cuda") for i in range(N)]
vit = graphed_clip_img_feat(clip_model)
GEN = dict(max_length=50, length_penalty=2.0, min_length=49, add_ner_ffn=True)

def inputs_fn(b):
    mask, _ = K.prep_ids(b["article_ids"], 1)
    nmask, _ = K.prep_ids(b["names_art_ids"], 1)
    return b["article_ids"], mask, vit(b["img_tensor"])[1], dict(face_features=b["face_emb"], face_mask=K.face_mask(b["face_emb"]),
                                                                 name_ids=b["names_art_ids"], name_mask=nmask)

with torch.no_grad():
    for rep in range(2):
        outs_seq = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in batches:
            src, mask, cls, kw = inputs_fn(b)
            outs_seq.append(model.generate(input_ids=src, attention_mask=mask, num_beams=5, image_features=cls, **kw, **GEN).cpu())
        torch.cuda.synchronize(); t_seq = (time.perf_counter() - t0) / N
    pipe = CaptionPipeline(model, inputs_fn, 5, **GEN)
    for rep in range(2):
        outs_pipe = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _, gen in pipe(batches):
            outs_pipe.append(gen.cpu())
        torch.cuda.synchronize(); t_pipe = (time.perf_counter() - t0) / N
same = all(torch.equal(a, b) for a, b in zip(outs_seq, outs_pipe))
print(json.dumps({"sequential_ms_per_caption": round(t_seq * 1e3, 2), "sequential_captions_per_s": round(1 / t_seq, 2),
                  "pipeline_ms_per_caption": round(t_pipe * 1e3, 2), "pipeline_captions_per_s": round(1 / t_pipe, 2), "ids_identical": same}))
