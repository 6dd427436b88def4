"""GPU box: the sensitive config-5 fixture, case 'plain', three generate() calls (eager / capture / replay): encoder output and ids per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cfg5_fixture as F5
from vacnic_amd import generate as Gn, kernels as K, synthetic
from vacnic_amd.config import ClipVisionConfig
from vacnic_amd.training import build_models

planted = np.load(F5.PLANTED_M4)
cfg = F5.cfg5_cfg()
vcfg = ClipVisionConfig(width=128, layers=1, patch_size=16, image_size=32, output_dim=64)
sd = F5.state_dict(cfg, planted)
model, _, _ = build_models(cfg, vcfg, init="synthetic", state_dicts=(sd, synthetic.make_state_dict(synthetic.guide_bart_param_shapes(cfg), seed=2),
                                                                      synthetic.make_state_dict(synthetic.clip_visual_param_shapes(vcfg), seed=4, std=0.05)))
model.eval()
batch, img = F5.inputs(cfg)
dev = {k: v.cuda() for k, v in batch.items()}
mask, _ = K.prep_ids(dev["article_ids"], 1)
nmask, _ = K.prep_ids(dev["names_art_ids"], 1)
encs = []
orig = Gn.GraphedCall.__call__
def spy(self, *a):
    o = orig(self, *a); torch.cuda.synchronize(); encs.append(o.float().clone()); return o
Gn.GraphedCall.__call__ = spy
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "generate_cfg5_m4.npz"))
if len(sys.argv) > 1:
    K._FIX_CAPTURE_FLOOR = int(sys.argv[1]) << 20
for name, extra in F5.CASES:
    want = torch.from_numpy(gold[name])
    for leg in ("eager", "capture", "replay", "host"):
        kw = {"device_beams": False} if leg == "host" else {}
        n0 = len(encs)
        out = model.generate(input_ids=dev["article_ids"], attention_mask=mask, num_beams=F5.NUM_BEAMS, max_length=F5.MAX_LENGTH, length_penalty=F5.LENGTH_PENALTY,
                             image_features=img.cuda(), face_features=dev["face_emb"], face_mask=K.face_mask(dev["face_emb"]),
                             name_ids=dev["names_art_ids"], name_mask=nmask, add_ner_ffn=True, **extra, **kw).cpu()
        d = (encs[-1] - encs[0]).abs()
        torch.cuda.synchronize()
        for st_, (ws_, cnt_) in K._FIX_CAPTURE.items():
            print(f"    capture-table {st_}: ws {ws_.data_ptr():#x} +{ws_.numel()}  counters {cnt_.data_ptr():#x} +{cnt_.numel() * 4}  non-zero counters {int((cnt_ != 0).sum())}")
        for st_, (ws_, cnt_) in K._FIX.items():
            print(f"    eager-table   stream {st_:#x}: ws {ws_.data_ptr():#x} +{ws_.numel()}  counters {cnt_.data_ptr():#x} +{cnt_.numel() * 4}  non-zero counters {int((cnt_ != 0).sum())}")
        a, b = out[0].tolist(), want[0].tolist()
        print(f"{name:8s} {leg:8s} encoder max |diff to first call| {float(d.max()):.4g}  ids == golden: {a == b}  first differing position "
              f"{next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)}", flush=True)
