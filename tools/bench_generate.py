"""BASELINE config 5: captions/sec on 1x MI355X — BART-large + ViT-L/14 full model, batch 1 (test_batch_size 1,
run_full_train.sh:10), beam 5, max_length 50, length_penalty 2.0, seed 42; synthetic GoodNews-shaped inputs."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import kernels as K, synthetic
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.models.clip_vit import extract_clip_img_feat, graphed_clip_img_feat
from vacnic_amd.training import build_models, to_device


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1 = BASELINE configs[4]; >1 = batched decode service (the reference never batches)
    cfg, vcfg = bart_large_vit_l14()
    model, _, clip_model = build_models(cfg, vcfg, device="cuda", seed=42, init="device")
    model.eval()
    times = []
    for i in range(n + 2):
        b = to_device(synthetic.make_batch(cfg, bsz, S=512, T=64, seed=42, step=i, full_length=True), "cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mask, _ = K.prep_ids(b["article_ids"], 1)
        nmask, _ = K.prep_ids(b["names_art_ids"], 1)
        _, cls = graphed_clip_img_feat(clip_model)(b["img_tensor"])
        out = model.generate(input_ids=b["article_ids"], attention_mask=mask, num_beams=5, max_length=50, length_penalty=2.0,
                             min_length=49,      # random-init weights emit EOS at once; force full-length captions (worst case)
                             image_features=cls, face_features=b["face_emb"], face_mask=K.face_mask(b["face_emb"]),
                             name_ids=b["names_art_ids"], name_mask=nmask, add_ner_ffn=True)
        torch.cuda.synchronize()
        if i >= 2:
            times.append(time.perf_counter() - t0)
    t = sum(times) / len(times)
    print(json.dumps({"metric": "captions/sec, beam 5, max_length 50, length_penalty 2.0" + (", batch 1 (BASELINE configs[4])" if bsz == 1 else f", batch {bsz} (batched decode service)"),
                      "value": round(bsz / t, 3), "unit": "captions/s", "batch": bsz, "ms_per_batch": round(t * 1e3, 1), "tokens": int(out.shape[1]),
                      "ms_per_token": round(t * 1e3 / out.shape[1], 2), "n": len(times), "data": "synthetic", "dtype": "bf16"}))


if __name__ == "__main__":
    main()
