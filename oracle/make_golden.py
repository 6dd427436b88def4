"""Generate tests/golden/*.npz from the REAL reference (runs only in the authoring container).

The reference Python at /root/reference is imported read-only (no bytecode written) with the shims
SURVEY §8c lists: a stub `clip` module, Tensor.cuda -> identity, argparse argv for the trainer, and
exec of the BatchSoftmax slice that lives inside the trainer's `__main__`.  Weights and batches come
from vacnic_amd.synthetic (seeded, name-keyed), so the tests regenerate identical inputs anywhere;
the fixtures hold only OUTPUTS of the reference (small slices, scalars, checksums).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import importlib
import os
import sys
import textwrap
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch

from vacnic_amd import synthetic
from vacnic_amd.config import ClipVisionConfig, VacnicConfig

MFULL_MOD = "src.models.modeling_mmbart_clip_inside_vis_clipcap_ent_type_final_fix_len_enc_self_face_name_ids_crossattn"
MVIS_MOD = "src.models.modeling_mmbart_clip_inside_vis_clipcap_ent_type_final_fix_len_enc_self_crossattn"
TRAIN_FILE = "train_mmbart_enc_self_face_name_ids_retrieve_crossattn_bart_guide_match.py"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.modules.setdefault("clip", types.ModuleType("clip"))
    torch.Tensor.cuda = lambda self, *a, **k: self          # hard-coded .cuda() in forward (MFULL:698,...)
    sys.path.insert(0, REF)
    mfull = importlib.import_module(MFULL_MOD)
    mvis = importlib.import_module(MVIS_MOD)
    argv = sys.argv
    sys.argv = ["train", "--enc_fusion_layer", "0", "--gpu_ids", "0"]
    spec = importlib.util.spec_from_file_location("ref_train", os.path.join(REF, TRAIN_FILE))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    sys.argv = argv
    train.torch = torch
    lines = open(os.path.join(REF, TRAIN_FILE)).read().split("\n")
    ns = {"torch": torch}
    exec(textwrap.dedent("\n".join(lines[630:660])), ns)       # batch_softmax + BatchSoftmax (TRAIN:631-660)
    return mfull, mvis, train, ns["BatchSoftmax"]


def bart_config(cfg: VacnicConfig):
    from transformers import BartConfig
    return BartConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, encoder_layers=cfg.encoder_layers,
                      decoder_layers=cfg.decoder_layers, encoder_attention_heads=cfg.encoder_attention_heads,
                      decoder_attention_heads=cfg.decoder_attention_heads, encoder_ffn_dim=cfg.encoder_ffn_dim,
                      decoder_ffn_dim=cfg.decoder_ffn_dim, max_position_embeddings=cfg.max_position_embeddings,
                      activation_function="gelu", scale_embedding=False, output_hidden_states=True, dropout=0.0,
                      attention_dropout=0.0, activation_dropout=0.0, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                      decoder_start_token_id=2, use_cache=False)


def build_ref_model(mod, cfg: VacnicConfig, sd):
    kw = dict(enc_fusion_layer=list(cfg.enc_fusion_layer), dim_common=cfg.dim_common, img_size=768,
              prompt_mlp_type=cfg.prompt_mlp_type, prompt_size=cfg.prompt_size, clip_model=None, max_ner_type_len=cfg.max_ner_type_len,
              max_ner_type_len_gt=cfg.max_ner_type_len_gt)
    if mod.__name__ == MFULL_MOD:
        kw["only_image"] = cfg.only_image
    if cfg.prompt_mlp_type == "mlp":
        kw["map_size"] = list(cfg.map_size)
    if cfg.init_attn_weight:
        kw["init_attn_weight"] = True
    m = mod.BartForMultiModalGeneration(bart_config(cfg), **kw)
    if cfg.clip_width != 768 and cfg.prompt_mlp_type == "clipcap":
        # documented one-line deviation (SURVEY "facts"): the 768 at MFULL:1136 is the CLIP width
        P = cfg.prompt_size
        m.model.encoder.prompt_mlp = mod.MLPClipCap((cfg.clip_width, (768 * P) // 2, 768 * P))
    m.lm_head.weight = m.model.shared.weight                  # transformers 5.x does not auto-tie here
    full = dict(sd)
    full["model.encoder.embed_tokens.weight"] = sd["model.shared.weight"]
    full["model.decoder.embed_tokens.weight"] = sd["model.shared.weight"]
    full["lm_head.weight"] = sd["model.shared.weight"]
    missing, unexpected = m.load_state_dict(full, strict=False)
    missing = [k for k in missing if k != "final_logits_bias"]
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval()


def small_cfg(**kw):
    base = dict(d_model=768, encoder_layers=2, decoder_layers=2, encoder_attention_heads=12, decoder_attention_heads=12,
                encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=768,
                dropout=0.0)
    base.update(kw)
    return VacnicConfig(**base)


def mlp_cfg():
    """--prompt_mlp_type mlp --map_size 12 32 16 8 (MFULL:76-108,1138): 12 patch tokens (not a multiple of 8, like ViT-B/16's 196)
    mixed down to an 8-token prompt, then visual_map 768 -> 1024."""
    return small_cfg(d_model=1024, encoder_layers=1, decoder_layers=1, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=2048, decoder_ffn_dim=2048, dim_common=1024, clip_width=768, prompt_mlp_type="mlp",
                     map_size=[12, 32, 16, 8])


def tied_cfg():
    """--init_attn_weight True (MFULL:1858-1870): name self-attn + image/name cross-attn weights tied to the text self-attn."""
    return small_cfg(encoder_layers=2, decoder_layers=1, enc_fusion_layer=[0, 1], init_attn_weight=True)


def slices(t, n=4096):
    """deterministic strided sample of a tensor + its full-tensor checksum."""
    f = t.detach().reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy(), np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()])


def run_full_case(name, mfull, train, BatchSoftmax, cfg, B, S, T, F, grads=True):
    torch.manual_seed(0)
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    if cfg.init_attn_weight:
        synthetic.apply_init_attn_weight(sd, cfg)
    gcfg = VacnicConfig(**{**cfg.__dict__, "enc_fusion_layer": [], "only_image": False, "init_attn_weight": False})
    sd_g = synthetic.make_state_dict(synthetic.guide_bart_param_shapes(cfg), seed=2)
    model = build_ref_model(mfull, cfg, sd)
    batch = synthetic.make_batch(cfg, B, S=S, T=T, F=F, seed=7, image_size=32)
    src, tgt = batch["article_ids"], batch["caption_ids"]
    tgt_in = train.shift_tokens_right(tgt, 1, 2)
    src_mask = train.create_src_mask_bart(src)
    face_mask = train.create_src_mask_bart(batch["face_emb"][:, :, -1])
    name_mask = train.create_src_mask_bart(batch["names_art_ids"])
    img_cls = synthetic.image_features(cfg, B)            # CLS vector (clipcap) or patch tokens (--prompt_mlp_type mlp)
    for p in model.parameters():
        p.requires_grad_(True)
    out = model(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in, image_features=img_cls,
                face_features=batch["face_emb"], face_mask=face_mask, name_ids=batch["names_art_ids"], name_mask=name_mask,
                add_ner_ffn=True)
    logits = out["logits"]
    txt = torch.nn.CrossEntropyLoss(ignore_index=1)(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1))
    # guide BART: vanilla arithmetic is the vendored MFULL class with no fusion layers (TRAIN:745)
    guide_sd = {k: v for k, v in sd_g.items()}
    for k, v in synthetic.make_state_dict(synthetic.mmbart_param_shapes(gcfg), seed=99).items():
        guide_sd.setdefault(k, v)                                   # unused VACNIC extras of the class
    guide = build_ref_model(mfull, gcfg, guide_sd)
    with torch.no_grad():
        gout = guide(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in, image_features=img_cls,
                     face_features=batch["face_emb"], face_mask=face_mask, name_ids=batch["names_art_ids"],
                     name_mask=name_mask, add_ner_ffn=True)
    gh = gout["decoder_hidden_states"][-1]
    tgt_mask = train.create_src_mask_bart(tgt)
    a = train.pool(out["decoder_hidden_states"][-1], tgt_mask); b = train.pool(gh, tgt_mask)
    a = a / a.norm(dim=1, keepdim=True); b = b / b.norm(dim=1, keepdim=True)
    colam = torch.nn.HingeEmbeddingLoss(margin=1.0)(torch.matmul(a, b.t()).diag(), -torch.ones(B))
    train.args.gpu_ids = "0"
    names = train.get_embedding_ner(model=model, ner_ids_3d=batch["names_ids"])
    secla = BatchSoftmax()(out["hidden_states_face"], names)
    loss = txt + 1.0 * secla + 0.5 * colam
    rec = {"txt": txt.item(), "colam": colam.item(), "secla": secla.item(), "loss": loss.item()}
    for key in ("logits", "hidden_states_face", "hidden_states_ner", "hidden_states_img", "encoder_last_hidden_state"):
        rec[key + "_s"], rec[key + "_c"] = slices(out[key])
    rec["dec_last_s"], rec["dec_last_c"] = slices(out["decoder_hidden_states"][-1])
    rec["guide_last_s"], rec["guide_last_c"] = slices(gh)
    rec["names_s"], rec["names_c"] = slices(names)
    rec["argmax"] = logits.argmax(-1).numpy()
    rec["lse"] = torch.logsumexp(logits, -1).detach().numpy()
    if grads:
        loss.backward()
        for pname in ("model.shared.weight", "model.encoder.layers.0.self_attn.q_proj.weight",
                      "model.encoder.layers.1.self_attn.out_proj.weight", "model.encoder.layers.0.self_attn.k_proj.weight",
                      "model.encoder.layers.1.cross_attn_img_ner.q_proj.bias", "model.encoder.layers.0.self_attn_img_name.v_proj.bias",
                      "model.encoder.layers.0.ner_map_up.weight", "model.encoder.layers.0._face_up.weight",
                      "model.encoder.layers.0.cross_attn_img_ner.k_proj.weight", "model.encoder.layers.1.fc1.weight",
                      "model.encoder.prompt_mlp.model.0.weight", "model.encoder.prompt_mlp.model.2.bias",
                      "model.encoder.prompt_mlp.model.4.weight", "model.encoder.prompt_mlp.model.4.bias",
                      "model.encoder._linear_1.weight", "model.encoder.embed_tokens_ner.weight",
                      "model.encoder.embed_positions.weight", "model.encoder.layernorm_embedding.weight",
                      "model.decoder.layers.1.encoder_attn.v_proj.weight", "model.decoder.layers.0.fc2.bias",
                      "model.decoder.layers.1.final_layer_norm.weight", "model.encoder.layers.0.img_layer_norm.bias",
                      "model.encoder.visual_map.weight"):
            params = dict(model.named_parameters())
            if pname in params and params[pname].grad is not None:
                rec["grad:" + pname + "_s"], rec["grad:" + pname + "_c"] = slices(params[pname].grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in rec.items()})
    print(name, {k: rec[k] for k in ("txt", "colam", "secla", "loss")})


def run_mvis_case(name, mvis, train, cfg, B, S, T):
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    model = build_ref_model(mvis, cfg, sd)
    batch = synthetic.make_batch(cfg, B, S=S, T=T, seed=8, image_size=32)
    src, tgt = batch["article_ids"], batch["caption_ids"]
    tgt_in = train.shift_tokens_right(tgt, 1, 2)
    src_mask = train.create_src_mask_bart(src)
    img_cls = synthetic._normal("img_cls", (B, cfg.clip_width), 1.0, 3)
    out = model(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in, image_features=img_cls)
    logits = out["logits"]
    txt = torch.nn.CrossEntropyLoss(ignore_index=1)(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1))
    rec = {"txt": txt.item()}
    rec["logits_s"], rec["logits_c"] = slices(logits)
    rec["hidden_states_img_s"], rec["hidden_states_img_c"] = slices(out["hidden_states_img"])
    rec["argmax"] = logits.argmax(-1).numpy()
    txt.backward()
    params = dict(model.named_parameters())
    for pname in ("model.shared.weight", "model.encoder.layers.0.cross_attn_img_ner.q_proj.weight",
                  "model.encoder.prompt_mlp.model.0.weight", "model.decoder.layers.0.self_attn.out_proj.bias"):
        rec["grad:" + pname + "_s"], rec["grad:" + pname + "_c"] = slices(params[pname].grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in rec.items()})
    print(name, rec["txt"])


GEN_CASES = [  # (num_beams, length_penalty, extra generate kwargs)
    (5, 2.0, {}), (5, 1.0, {}), (1, 1.0, dict(min_length=4)), (5, 1.0, dict(min_length=4)),
    (4, 2.0, dict(min_length=3, no_repeat_ngram_size=2)), (3, 0.5, dict(min_length=5, no_repeat_ngram_size=3, early_stopping=True)),
             # the generation defaults of the facebook/bart-* hub checkpoints (config.HUB_GENERATION_DEFAULTS) at config 5's beam / length penalty
             (5, 2.0, dict(no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0))]


def run_generate_case(mfull, train):
    """Beam search of transformers 5.15 (GenerationMixin, cache-less) over the REFERENCE model's logits: pins the
    oracle's restatement of the 4.18 beam-search bookkeeping (generate lives in the un-vendored HF package)."""
    from transformers import GenerationMixin
    cfg = small_cfg(encoder_layers=1, decoder_layers=1)
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    # N(0, 0.02) weights give almost flat next-token distributions (top-2 margins far below bf16 resolution, so any two
    # correct implementations order the beams differently); scaling the tied embedding/LM-head matrix spreads the logits
    # (std ~3) the way a trained model does, which makes "identical ids" a meaningful, stable bar
    sd["model.shared.weight"] = sd["model.shared.weight"] * synthetic.GEN_SHARPEN

    class Oracle(mfull.BartForMultiModalGeneration, GenerationMixin):
        pass
    orig = mfull.BartForMultiModalGeneration
    mfull.BartForMultiModalGeneration = Oracle
    m = build_ref_model(mfull, cfg, sd)
    mfull.BartForMultiModalGeneration = orig
    batch = synthetic.make_batch(cfg, 2, S=24, T=8, F=2, seed=9, image_size=32)
    src = batch["article_ids"]; mask = train.create_src_mask_bart(src)
    img = synthetic._normal("img_cls", (2, 768), 1.0, 3)
    kw = dict(image_features=img, face_features=batch["face_emb"], face_mask=train.create_src_mask_bart(batch["face_emb"][:, :, -1]),
              name_ids=batch["names_art_ids"], name_mask=train.create_src_mask_bart(batch["names_art_ids"]), add_ner_ffn=True)
    rec = {}
    for i, (nb, lp, extra) in enumerate(GEN_CASES):
        out = m.generate(input_ids=src, attention_mask=mask, num_beams=nb, max_length=12, length_penalty=lp, use_cache=False,
                         do_sample=False, **extra, **kw).sequences
        rec[f"seq{i}"] = out.numpy()
        print("generate", nb, lp, extra, out.tolist())
    np.savez_compressed(os.path.join(OUT, "generate_small.npz"), **rec)


def run_generate_cfg5_case(mfull, train):
    """configs[4]'s exact call — batch 1, num_beams 5, max_length 50, length_penalty 2.0 (TRAIN:513-520, DDPINF:758-842,
    run_full_train.sh:10-11) — by transformers' beam search (GenerationMixin, cache-less) over the REFERENCE model carrying the
    planted-caption weights of oracle/cfg5_fixture.py: with the library defaults, with the hub checkpoints' generation defaults,
    and each with min_length 49 so that every one of the 50 positions is decoded.  Fixture: tests/golden/generate_cfg5.npz."""
    from transformers import GenerationMixin
    from oracle import cfg5_fixture as F5
    cfg = F5.cfg5_cfg()
    sd = F5.state_dict(cfg)
    sd_m4 = F5.state_dict(cfg, np.load(F5.PLANTED_M4))

    class Oracle(mfull.BartForMultiModalGeneration, GenerationMixin):
        pass
    orig = mfull.BartForMultiModalGeneration
    mfull.BartForMultiModalGeneration = Oracle
    m = build_ref_model(mfull, cfg, sd)
    mfull.BartForMultiModalGeneration = orig
    batch, img = F5.inputs(cfg)
    src = batch["article_ids"]; mask = train.create_src_mask_bart(src)
    kw = dict(image_features=img, face_features=batch["face_emb"], face_mask=train.create_src_mask_bart(batch["face_emb"][:, :, -1]),
              name_ids=batch["names_art_ids"], name_mask=train.create_src_mask_bart(batch["names_art_ids"]), add_ner_ffn=True)
    rec = {}
    for name, extra in F5.CASES:
        out = m.generate(input_ids=src, attention_mask=mask, num_beams=F5.NUM_BEAMS, max_length=F5.MAX_LENGTH,
                         length_penalty=F5.LENGTH_PENALTY, use_cache=False, do_sample=False, **extra, **kw).sequences
        rec[name] = out.numpy()
        print("generate cfg5", name, tuple(out.shape), out[0, :8].tolist(), "...", out[0, -3:].tolist())
    np.savez_compressed(os.path.join(OUT, "generate_cfg5.npz"), **rec)
    # ---- the sensitive variant (cfg5_fixture "m4": 4 .. 7-unit margins, n-gram trap): best sequence AND the whole n-best list
    # (num_return_sequences = num_beams: sequences + sequences_scores of BeamSearchScorer.finalize) of the reference
    mfull.BartForMultiModalGeneration = Oracle
    m4 = build_ref_model(mfull, cfg, sd_m4)
    mfull.BartForMultiModalGeneration = orig
    rec = {}
    for name, extra in F5.CASES:
        o = m4.generate(input_ids=src, attention_mask=mask, num_beams=F5.NUM_BEAMS, max_length=F5.MAX_LENGTH,
                        length_penalty=F5.LENGTH_PENALTY, use_cache=False, do_sample=False, num_return_sequences=F5.NUM_BEAMS,
                        output_scores=True, return_dict_in_generate=True, **extra, **kw)
        seqs = o.sequences.numpy()
        rec[name] = seqs[:1, :int((seqs[0] != 1).sum())]                  # best hypothesis without its padding
        rec[name + "_nbest"] = seqs
        rec[name + "_nbest_scores"] = o.sequences_scores.numpy().astype(np.float64)
        print("generate cfg5 m4", name, tuple(seqs.shape), "scores", [round(float(v), 5) for v in o.sequences_scores], "lens", (seqs != 1).sum(1).tolist())
    # without the ban the reference walks into the n-gram trap (the fixture's sensitivity claim, pinned by the reference itself)
    o = m4.generate(input_ids=src, attention_mask=mask, num_beams=F5.NUM_BEAMS, max_length=F5.MAX_LENGTH, length_penalty=F5.LENGTH_PENALTY,
                    use_cache=False, do_sample=False, early_stopping=True, forced_bos_token_id=0, **kw).sequences
    rec["hub_without_ngram_ban"] = o.numpy()
    np.savez_compressed(os.path.join(OUT, "generate_cfg5_m4.npz"), **rec)


def run_helpers(train, BatchSoftmax):
    g = torch.Generator().manual_seed(5)
    ids = torch.tensor([[0, 5, 6, 2, 1], [0, 9, 2, 1, 1], [0, 7, -100, 2, 1]])
    rec = {"shift": train.shift_tokens_right(ids, 1, 2).numpy(), "mask": train.create_src_mask_bart(ids).numpy()}
    h = torch.randn(4, 6, 16, generator=g)
    m = torch.tensor([[1, 1, 1, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1]])
    rec["pool_in"] = h.numpy(); rec["pool_mask"] = m.numpy(); rec["pool_out"] = train.pool(h, m).numpy()
    face = torch.randn(5, 3, 32, generator=g); ner = torch.randn(5, 4, 32, generator=g)
    rec["secla_face"] = face.numpy(); rec["secla_ner"] = ner.numpy(); rec["secla_out"] = BatchSoftmax()(face, ner).item()
    x = torch.randn(7, generator=g)
    rec["hinge_in"] = x.numpy()
    rec["hinge_out"] = np.array([torch.nn.HingeEmbeddingLoss(margin=mg)(x, -torch.ones(7)).item() for mg in (1.0, 0.3)])
    # AdamW + get_linear_schedule_with_warmup (transformers) — 6 steps
    from transformers import get_linear_schedule_with_warmup
    p = torch.nn.Parameter(torch.randn(64, generator=g))
    rec["adam_p0"] = p.detach().clone().numpy()
    gs = torch.randn(6, 64, generator=g)
    rec["adam_g"] = gs.numpy()
    opt = torch.optim.AdamW([p], betas=(0.9, 0.999), lr=3e-5, eps=1e-8, weight_decay=0.01)
    sch = get_linear_schedule_with_warmup(opt, 2.0, 40.0)
    lrs = []
    for i in range(6):
        p.grad = gs[i].clone()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step(); opt.zero_grad()
    rec["adam_p6"] = p.detach().numpy(); rec["adam_lrs"] = np.array(lrs)
    np.savez_compressed(os.path.join(OUT, "trainer_helpers.npz"), **rec)
    print("helpers ok")


def run_clip_crosscheck():
    """openai-CLIP is not installed: cross-check the oracle ViT restatement against transformers.CLIPVisionModel
    (same architecture: pre-LN, QuickGELU, class token, ln_pre/ln_post) with mapped weights; store its output."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    v = ClipVisionConfig(width=128, layers=2, patch_size=16, image_size=64, output_dim=64)
    sd = synthetic.make_state_dict(synthetic.clip_visual_param_shapes(v), seed=4, std=0.05)
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                                          image_size=64, patch_size=16, hidden_act="quick_gelu", layer_norm_eps=1e-5)).eval()
    m = {}
    vm = "" if not any(k.startswith("vision_model.") for k in hf.state_dict()) else "vision_model."
    m[vm + "embeddings.class_embedding"] = sd["class_embedding"]
    m[vm + "embeddings.patch_embedding.weight"] = sd["conv1.weight"]
    m[vm + "embeddings.position_embedding.weight"] = sd["positional_embedding"]
    for a, b in (("pre_layrnorm", "ln_pre"), ("post_layernorm", "ln_post")):
        m[vm + a + ".weight"] = sd[b + ".weight"]; m[vm + a + ".bias"] = sd[b + ".bias"]
    for i in range(2):
        L = f"transformer.resblocks.{i}"; Hf = f"{vm}encoder.layers.{i}"
        w, bb = sd[L + ".attn.in_proj_weight"], sd[L + ".attn.in_proj_bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            m[f"{Hf}.self_attn.{nm}.weight"] = w[j * 128:(j + 1) * 128]; m[f"{Hf}.self_attn.{nm}.bias"] = bb[j * 128:(j + 1) * 128]
        m[f"{Hf}.self_attn.out_proj.weight"] = sd[L + ".attn.out_proj.weight"]; m[f"{Hf}.self_attn.out_proj.bias"] = sd[L + ".attn.out_proj.bias"]
        for a, b in (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            m[f"{Hf}.{a}.weight"] = sd[f"{L}.{b}.weight"]; m[f"{Hf}.{a}.bias"] = sd[f"{L}.{b}.bias"]
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    img = synthetic._normal("clip_img", (2, 3, 64, 64), 1.0, 5)
    with torch.no_grad():
        o = hf(pixel_values=img)
    np.savez_compressed(os.path.join(OUT, "clip_vit_hf.npz"), x_cls=o.pooler_output.numpy(),
                        last_hidden=o.last_hidden_state.numpy())
    print("clip ok", o.pooler_output.abs().mean().item())


COLLATE_CASES = {          # name -> (seed, batch size, forced per-sample edits)
    "mixed": (11, 6, None),
    "no_faces_anywhere": (12, 3, "nofaces"),
    "single_name_row": (13, 4, "noname"),
}
DSG_FILE = "src/data/goodnews_dataset_entity_type_newsmep_ent_ent_pos.py"


def collate_case_samples(case):
    seed, n, edit = COLLATE_CASES[case]
    samples = synthetic.make_samples(n, seed=seed)
    for i, sm in enumerate(samples):
        if edit == "nofaces":
            sm["face_emb"] = sm["face_emb"][:0]
        if edit == "noname":                               # every sample carries only the <NONAME> row: max_num_seq == 1 branch
            sm["names_ids"] = np.array([[0, 50266, 2]], dtype=np.int64)
    return samples


def run_collate():
    """The reference's collate_fn_goodnews_entity_type + its padding helpers (DSG:22-305, pure torch/python; the module itself
    imports spacy/unidecode at the top, so the slice is exec'd like BatchSoftmax) over seeded synthetic per-sample records."""
    src = open(os.path.join(REF, DSG_FILE)).read()
    a = src.index("def collate_fn_goodnews_entity_type(batch):")
    b = src.index("def make_new_entity_ids(")
    ns = {"torch": torch}
    exec(compile(src[a:b], "dsg_slice", "exec"), ns)
    rec = {}
    for case in COLLATE_CASES:
        samples = collate_case_samples(case)
        batch = []
        for sm in samples:
            f = sm["face_emb"].astype(np.float32)
            batch.append({
                "article": "", "article_ids": torch.from_numpy(sm["article_ids"])[None], "article_ner_mask_ids": torch.zeros((1, 1), dtype=torch.long),
                "caption": "", "caption_ids": torch.from_numpy(sm["caption_ids"])[None], "caption_ids_clip": None,
                "names_art": [], "org_norp_gpe_loc_art": [], "names_art_ids": torch.from_numpy(sm["names_art_ids"])[None],
                "org_norp_gpe_loc_art_ids": torch.from_numpy(sm["names_art_ids"])[None], "names": [], "org_norp_gpe_loc": [],
                "names_ids": sm["names_ids"].tolist(), "org_norp_gpe_loc_ids": sm["names_ids"].tolist(),
                "all_gt_ner_ids": torch.zeros((1, 4), dtype=torch.long), "all_gt_ner": [],
                "face_emb": f if len(f) else np.zeros((1, 0), dtype=np.float32),       # the dataset yields an empty 2-D array when no face was detected
                "obj_emb": np.zeros((1, 0), dtype=np.float32), "img_tensor": torch.from_numpy(sm["image"].astype(np.float32))[None],
                "person_id_positions": [], "person_id_positions_cap": [],
                "names_ids_flatten": torch.from_numpy(sm["names_ids_flatten"])[None],
                "org_norp_gpe_loc_ids_flatten": torch.from_numpy(sm["names_ids_flatten"])[None]})
        out = ns["collate_fn_goodnews_entity_type"](batch)
        for k in ("article_ids", "caption_ids", "names_art_ids", "names_ids", "names_ids_flatten", "face_emb", "img_tensor"):
            rec[f"{case}:{k}"] = out[k].numpy()
        print("collate", case, {k: tuple(out[k].shape) for k in ("article_ids", "caption_ids", "names_ids", "face_emb")})
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **rec)


def run_checkpoint_readback(mfull, train):
    """SURVEY §8f-3 "can be read back by the reference": a checkpoint written by vacnic_amd/checkpoint.py (oracle/ckpt_case.py)
    is loaded into the REAL reference class with load_state_dict(strict=True) — no renaming, no missing or unexpected key —
    and the reference's teacher-forced logits on a seeded batch are recorded."""
    from oracle import ckpt_case
    cfg, ck, batch, img = ckpt_case.make_checkpoint()
    m = build_ref_model(mfull, cfg, {k: v for k, v in ck["model"].items() if k in synthetic.mmbart_param_shapes(cfg)})
    fresh = mfull.BartForMultiModalGeneration(bart_config(cfg), enc_fusion_layer=list(cfg.enc_fusion_layer), dim_common=cfg.dim_common,
                                              img_size=768, prompt_mlp_type=cfg.prompt_mlp_type, prompt_size=cfg.prompt_size, clip_model=None,
                                              max_ner_type_len=cfg.max_ner_type_len, max_ner_type_len_gt=cfg.max_ner_type_len_gt,
                                              only_image=cfg.only_image)
    res = fresh.load_state_dict(ck["model"], strict=True)          # the checkpoint AS WRITTEN, into an untouched reference instance
    assert not res.missing_keys and not res.unexpected_keys, res
    fresh.eval()
    src, tgt = batch["article_ids"], batch["caption_ids"]
    kw = dict(face_features=batch["face_emb"], face_mask=train.create_src_mask_bart(batch["face_emb"][:, :, -1]),
              name_ids=batch["names_art_ids"], name_mask=train.create_src_mask_bart(batch["names_art_ids"]), add_ner_ffn=True)
    with torch.no_grad():
        outs = [mm(input_ids=src, attention_mask=train.create_src_mask_bart(src), decoder_input_ids=train.shift_tokens_right(tgt, 1, 2),
                   image_features=img, **kw)["logits"] for mm in (fresh, m)]
    assert torch.equal(outs[0], outs[1]), "strict load of the checkpoint == explicit tie + named load"
    rec = {"n_keys": np.array(len(ck["model"])), "step": np.array(ck["meta"]["step"]), "argmax": outs[0].argmax(-1).numpy()}
    rec["logits_s"], rec["logits_c"] = slices(outs[0])
    np.savez_compressed(os.path.join(OUT, "checkpoint_readback.npz"), **rec)
    print("checkpoint readback ok:", len(ck["model"]), "tensors, logits checksum", rec["logits_c"])


def main():
    os.makedirs(OUT, exist_ok=True)
    mfull, mvis, train, BatchSoftmax = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "checkpoint":
        run_checkpoint_readback(mfull, train)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "generate":
        run_generate_case(mfull, train)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "generate_cfg5":
        run_generate_cfg5_case(mfull, train)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "collate":
        run_collate()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mlp":
        run_full_case("mfull_mlp_d1024", mfull, train, BatchSoftmax, mlp_cfg(), B=2, S=40, T=10, F=2)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "tied":
        run_full_case("mfull_tied_d768", mfull, train, BatchSoftmax, tied_cfg(), B=2, S=40, T=10, F=2)
        return
    run_collate()
    run_helpers(train, BatchSoftmax)
    run_checkpoint_readback(mfull, train)
    run_generate_case(mfull, train)
    run_generate_cfg5_case(mfull, train)
    run_clip_crosscheck()
    run_full_case("mfull_d768", mfull, train, BatchSoftmax, small_cfg(), B=3, S=48, T=12, F=3)
    run_full_case("mfull_d1024", mfull, train, BatchSoftmax,
                  small_cfg(d_model=1024, encoder_layers=1, decoder_layers=1, encoder_attention_heads=16,
                            decoder_attention_heads=16, encoder_ffn_dim=2048, decoder_ffn_dim=2048, dim_common=1024,
                            clip_width=1024), B=2, S=40, T=10, F=2)
    run_mvis_case("mvis_d768", mvis, train, small_cfg(only_image=True, enc_fusion_layer=[0, 1]), B=2, S=32, T=8)
    run_full_case("mfull_mlp_d1024", mfull, train, BatchSoftmax, mlp_cfg(), B=2, S=40, T=10, F=2)
    run_full_case("mfull_tied_d768", mfull, train, BatchSoftmax, tied_cfg(), B=2, S=40, T=10, F=2)


if __name__ == "__main__":
    main()
