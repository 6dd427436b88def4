"""Diagnostic: is beam-search generation reproducible (a) run to run on one model, (b) across a checkpoint round trip into a
freshly built model?  Prints the first differing position per caption."""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import checkpoint, synthetic, streams
from vacnic_amd.config import ClipVisionConfig, VacnicConfig
from vacnic_amd.training import FusedAdamW, TrainArgs, build_models, gen_caption_from_loader_bart, to_device, train_step

cfg = VacnicConfig(d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12, decoder_attention_heads=12,
                   encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0, 1], dim_common=768, clip_width=768).validate()
vcfg = ClipVisionConfig(width=768, layers=12, patch_size=32, output_dim=512)


def batches(n=4):
    return [synthetic.make_batch(cfg, 1, S=64, T=64, seed=123, rank=0, step=i) for i in range(n)]


def diff(a, b, tag):
    for k in a:
        x, y = a[k]["gen"][0], b[k]["gen"][0]
        d = next((i for i, (p, q) in enumerate(zip(x, y)) if p != q), None if len(x) == len(y) else min(len(x), len(y)))
        print(tag, k, "same" if d is None else f"differs at {d}: {x[d:d+3]} vs {y[d:d+3]}")


streams.enable(True)
model, guide, _ = build_models(cfg, vcfg, seed=684331, init="device")
args = TrainArgs(num_training_steps=60, warmup_rate=0.05)
opt = FusedAdamW(model.arena, lr=3e-5, num_warmup_steps=3, num_training_steps=60)
for i in range(3):
    train_step(model, guide, opt, to_device(synthetic.make_batch(cfg, 2, S=64, T=64, seed=5, step=i), "cuda"), args)
torch.cuda.synchronize()
g1 = gen_caption_from_loader_bart(model, batches(), 2, 8)
g2 = gen_caption_from_loader_bart(model, batches(), 2, 8)
diff(g1, g2, "same model, second pass:")
buf = io.BytesIO(); checkpoint.save_checkpoint(buf, model, opt, step=3); buf.seek(0)
m2, _, _ = build_models(cfg, vcfg, seed=684331, init="device", with_guide=False)
checkpoint.load_checkpoint(buf, m2)
print("fp32 equal:", torch.equal(m2.arena.flat32, model.arena.flat32), "bf16 equal:", torch.equal(m2.arena.flat16, model.arena.flat16),
      "clip equal:", torch.equal(m2.clip_model.visual.arena.flat16, model.clip_model.visual.arena.flat16))
streams.enable(False)
g3 = gen_caption_from_loader_bart(m2, batches(), 2, 8)
diff(g1, g3, "restored model:")
g4 = gen_caption_from_loader_bart(m2, batches(), 2, 8)
diff(g3, g4, "restored model, second pass:")

# ---- bisect: encoder output, eager decode, graph decode on the differing batches
from vacnic_amd.training import _model_inputs
from vacnic_amd.generate import generate
model.eval(); m2.eval()
for bi, b in enumerate(batches()):
    b = to_device(b, "cuda")
    outs = []
    for net in (model, m2):
        src, mask, feats, kw = _model_inputs(net, b)
        enc = net.model.encoder(input_ids=src, attention_mask=mask, image_features=feats, **kw)
        outs.append((feats, enc["last_hidden_state"]))
    print(bi, "feats equal:", torch.equal(outs[0][0], outs[1][0]), "encoder equal:", torch.equal(outs[0][1], outs[1][1]))
    res = {}
    for name, net in (("orig", model), ("restored", m2)):
        src, mask, feats, kw = _model_inputs(net, b)
        for ug in (False, True):
            net.__dict__.pop("_decode_sessions", None)
            seqs = [generate(net, src, mask, num_beams=2, max_length=8, image_features=feats, use_graphs=ug, **kw).tolist() for _ in range(3)]
            res[(name, ug)] = seqs
    base = res[("orig", False)][0]
    for k, v in res.items():
        print(bi, k, ["same" if s == base else s for s in v])
