"""Seeded synthetic weights and GoodNews-shaped batches.  numpy Generators keyed by
parameter NAME, so the oracle, the HIP model and the fixture generator (which runs next to the
reference, in another container) all see bit-identical fp32 values without sharing any file."""
import zlib

import numpy as np
import torch

GEN_SHARPEN = 6.0     # scale of the tied embedding/LM-head matrix in the caption-generation test fixtures (spreads the logits like a trained model)

from .config import ClipVisionConfig, VacnicConfig


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def _normal(name, shape, std, seed):
    return torch.from_numpy((_rng(name, seed).standard_normal(shape, dtype=np.float32) * std).astype(np.float32))


def _attn_names(prefix):
    out = []
    for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
        out += [(f"{prefix}.{p}.weight", "w"), (f"{prefix}.{p}.bias", "b")]
    return out


def apply_init_attn_weight(sd, cfg: VacnicConfig):
    """name-keyed view of `--init_attn_weight True` (MFULL:1858-1870): the tied attentions' weight entries become the SAME tensor
    objects as the text self-attention's (so autograd on the dict accumulates the three uses into one gradient)."""
    for i in range(cfg.encoder_layers):
        for tied in ("self_attn_img_name", "cross_attn_img_ner"):
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[f"model.encoder.layers.{i}.{tied}.{proj}.weight"] = sd[f"model.encoder.layers.{i}.self_attn.{proj}.weight"]
    return sd


def image_features(cfg: VacnicConfig, B, seed=3):
    """stand-in for extract_clip_img_feat's output that feeds `image_features`: the ln_post CLS vector [B, clip_width]
    (ClipCap prompt) or the ln_post patch tokens [B, map_size[0], 768] (--prompt_mlp_type mlp)."""
    if cfg.prompt_mlp_type == "mlp":
        return _normal("img_feat", (B, cfg.map_size[0], 768), 1.0, seed)
    return _normal("img_cls", (B, cfg.clip_width), 1.0, seed)


def mmbart_param_shapes(cfg: VacnicConfig):
    """name -> shape, reference parameter names (MFULL state_dict order is not required)."""
    d, V = cfg.d_model, cfg.vocab_size
    s = {"model.shared.weight": (V, d)}
    e = "model.encoder"
    s[f"{e}.embed_positions.weight"] = (cfg.max_position_embeddings + 2, d)
    s[f"{e}.layernorm_embedding.weight"] = (d,); s[f"{e}.layernorm_embedding.bias"] = (d,)
    P = cfg.prompt_size
    if cfg.prompt_mlp_type == "mlp":               # MFULL:76-108: Linear at Sequential slots 0, 2, 4, ...
        for i in range(len(cfg.map_size) - 1):
            s[f"{e}.prompt_mlp.model.{2 * i}.weight"] = (cfg.map_size[i + 1], cfg.map_size[i])
            s[f"{e}.prompt_mlp.model.{2 * i}.bias"] = (cfg.map_size[i + 1],)
    else:
        s[f"{e}.prompt_mlp.model.0.weight"] = (768 * P // 2, cfg.clip_width); s[f"{e}.prompt_mlp.model.0.bias"] = (768 * P // 2,)
        s[f"{e}.prompt_mlp.model.2.weight"] = (768 * P, 768 * P // 2); s[f"{e}.prompt_mlp.model.2.bias"] = (768 * P,)
    if d == 1024:
        s[f"{e}.visual_map.weight"] = (1024, 768); s[f"{e}.visual_map.bias"] = (1024,)
    if not cfg.only_image:
        s[f"{e}.embed_tokens_ner.weight"] = (50267, d)
        s[f"{e}.embed_positions_ner.weight"] = (cfg.max_position_embeddings + 2, d)
        s[f"{e}.layernorm_embedding_ner.weight"] = (d,); s[f"{e}.layernorm_embedding_ner.bias"] = (d,)
    s[f"{e}._linear_1.weight"] = (cfg.dim_common, cfg.face_dim); s[f"{e}._linear_1.bias"] = (cfg.dim_common,)

    def attn(prefix):
        for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[f"{prefix}.{p}.weight"] = (d, d); s[f"{prefix}.{p}.bias"] = (d,)

    def ln(prefix):
        s[f"{prefix}.weight"] = (d,); s[f"{prefix}.bias"] = (d,)

    def lin(prefix, o, i):
        s[f"{prefix}.weight"] = (o, i); s[f"{prefix}.bias"] = (o,)

    for i in range(cfg.encoder_layers):
        L = f"{e}.layers.{i}"
        attn(f"{L}.self_attn"); ln(f"{L}.self_attn_layer_norm")
        lin(f"{L}.fc1", cfg.encoder_ffn_dim, d); lin(f"{L}.fc2", d, cfg.encoder_ffn_dim); ln(f"{L}.final_layer_norm")
        lin(f"{L}._linear_1up", cfg.encoder_ffn_dim, d); lin(f"{L}._linear_1down", d, cfg.encoder_ffn_dim); ln(f"{L}.img_layer_norm")
        if not cfg.only_image:
            lin(f"{L}.ner_map_up", 4 * cfg.max_ner_type_len_gt, cfg.max_ner_type_len)
            lin(f"{L}.ner_map_down", cfg.max_ner_type_len_gt, 4 * cfg.max_ner_type_len_gt); ln(f"{L}.ner_map_layer_norm")
            attn(f"{L}.self_attn_img_name"); ln(f"{L}.img_name_attn_layer_norm")
            lin(f"{L}._face_up", 3072, d); lin(f"{L}._face_down", d, 3072); ln(f"{L}.face_layer_norm")
        # MVIS keeps cross_attn_img_ner in the only-image model too (MVIS:560-590)
        attn(f"{L}.cross_attn_img_ner"); ln(f"{L}.img_ner_attn_layer_norm")
    dd = "model.decoder"
    s[f"{dd}.embed_positions.weight"] = (cfg.max_position_embeddings + 2, d)
    ln(f"{dd}.layernorm_embedding")
    for i in range(cfg.decoder_layers):
        L = f"{dd}.layers.{i}"
        attn(f"{L}.self_attn"); ln(f"{L}.self_attn_layer_norm")
        attn(f"{L}.encoder_attn"); ln(f"{L}.encoder_attn_layer_norm")
        lin(f"{L}.fc1", cfg.decoder_ffn_dim, d); lin(f"{L}.fc2", d, cfg.decoder_ffn_dim); ln(f"{L}.final_layer_norm")
    return s


def guide_bart_param_shapes(cfg: VacnicConfig):
    """HF BartForConditionalGeneration names (vanilla encoder/decoder; TRAIN:745)."""
    d, V = cfg.d_model, cfg.vocab_size
    s = {"model.shared.weight": (V, d)}
    for side, n, ffn in (("encoder", cfg.encoder_layers, cfg.encoder_ffn_dim), ("decoder", cfg.decoder_layers, cfg.decoder_ffn_dim)):
        p = f"model.{side}"
        s[f"{p}.embed_positions.weight"] = (cfg.max_position_embeddings + 2, d)
        s[f"{p}.layernorm_embedding.weight"] = (d,); s[f"{p}.layernorm_embedding.bias"] = (d,)
        for i in range(n):
            L = f"{p}.layers.{i}"
            attns = ["self_attn"] + (["encoder_attn"] if side == "decoder" else [])
            for a in attns:
                for q in ("k_proj", "v_proj", "q_proj", "out_proj"):
                    s[f"{L}.{a}.{q}.weight"] = (d, d); s[f"{L}.{a}.{q}.bias"] = (d,)
                s[f"{L}.{a}_layer_norm.weight"] = (d,); s[f"{L}.{a}_layer_norm.bias"] = (d,)
            s[f"{L}.fc1.weight"] = (ffn, d); s[f"{L}.fc1.bias"] = (ffn,)
            s[f"{L}.fc2.weight"] = (d, ffn); s[f"{L}.fc2.bias"] = (d,)
            s[f"{L}.final_layer_norm.weight"] = (d,); s[f"{L}.final_layer_norm.bias"] = (d,)
    return s


def clip_visual_param_shapes(v: ClipVisionConfig):
    """openai-CLIP VisionTransformer names (clip/model.py, `visual.` prefix stripped)."""
    w = v.width
    s = {"conv1.weight": (w, 3, v.patch_size, v.patch_size), "class_embedding": (w,),
         "positional_embedding": (v.tokens, w), "ln_pre.weight": (w,), "ln_pre.bias": (w,),
         "ln_post.weight": (w,), "ln_post.bias": (w,), "proj": (w, v.output_dim)}
    for i in range(v.layers):
        L = f"transformer.resblocks.{i}"
        s[f"{L}.attn.in_proj_weight"] = (3 * w, w); s[f"{L}.attn.in_proj_bias"] = (3 * w,)
        s[f"{L}.attn.out_proj.weight"] = (w, w); s[f"{L}.attn.out_proj.bias"] = (w,)
        s[f"{L}.ln_1.weight"] = (w,); s[f"{L}.ln_1.bias"] = (w,); s[f"{L}.ln_2.weight"] = (w,); s[f"{L}.ln_2.bias"] = (w,)
        s[f"{L}.mlp.c_fc.weight"] = (4 * w, w); s[f"{L}.mlp.c_fc.bias"] = (4 * w,)
        s[f"{L}.mlp.c_proj.weight"] = (w, 4 * w); s[f"{L}.mlp.c_proj.bias"] = (w,)
    return s


def make_state_dict(shapes, seed=0, std=0.02, perturb_norm=True, bias_std=0.02):
    """N(0, std^2) Linear/Embedding weights as BartPretrainedModel._init_weights (MFULL:899-908).
    LayerNorm gamma/beta and biases get small seeded perturbations (perturb_norm) so that parity tests
    exercise them (the reference init of 1/0/0 would hide a swapped gamma/beta or a dropped bias)."""
    sd = {}
    for name, shape in shapes.items():
        is_ln = ("layer_norm" in name or "layernorm" in name or name.startswith("ln_") or ".ln_" in name)
        if is_ln and name.endswith("weight"):
            t = 1.0 + (_normal(name, shape, 0.05, seed) if perturb_norm else torch.zeros(shape))
        elif name.endswith("bias"):
            t = _normal(name, shape, bias_std if perturb_norm else 0.0, seed)
        else:
            t = _normal(name, shape, std, seed)
        sd[name] = t
    return sd


def make_batch(cfg: VacnicConfig, B, S=512, T=64, F=4, Nn=5, Ln=8, seed=42, rank=0, step=0, full_length=False,
               image_size=224):
    """GoodNews-shaped synthetic batch, layout of collate_fn_goodnews_entity_type (DSG:22-127)."""
    g = np.random.default_rng([seed, rank, step])
    pad, bos, eos = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id

    def seq(n, L, lo, hi_len=None):
        out = np.full((n, L), pad, dtype=np.int64)
        for i in range(n):
            ln = L if full_length else int(g.integers(lo, (hi_len or L) + 1))
            ln = max(ln, 2)
            out[i, 0] = bos
            out[i, 1:ln - 1] = g.integers(3, 50265, size=ln - 2)
            out[i, ln - 1] = eos
        return out

    article = seq(B, S, max(2, S // 4))
    caption = seq(B, T, min(8, T))
    names_art = seq(B, cfg.max_ner_type_len, 2)
    names_ids = np.full((B, Nn, Ln), pad, dtype=np.int64)
    for b in range(B):
        for i in range(Nn - 1):
            k = int(g.integers(1, Ln - 1))
            names_ids[b, i, 0] = bos; names_ids[b, i, 1:1 + k] = g.integers(3, 50265, size=k); names_ids[b, i, 1 + k] = eos
        names_ids[b, Nn - 1, :3] = [bos, 50266, eos]            # <NONAME> row (DSG:111)
    img = g.standard_normal((B, 3, image_size, image_size), dtype=np.float32)
    face = (g.standard_normal((B, F, cfg.face_dim), dtype=np.float32) / np.sqrt(cfg.face_dim)).astype(np.float32)
    for b in range(B):
        nf = F if full_length else int(g.integers(0, F + 1))
        face[b, nf:] = 1.0                                       # pad faces are all-ones rows (DSG:48,124)
    t = torch.from_numpy
    return {"article_ids": t(article), "caption_ids": t(caption), "img_tensor": t(img), "face_emb": t(face),
            "names_art_ids": t(names_art), "names_ids": t(names_ids)}


def make_samples(n, seed=0, max_article=96, max_caption=24, max_faces=4, max_names=4, image_size=32, ner_len=80, gt_len=20,
                 pad=1, bos=0, eos=2, ent=50265, noname=50266, vocab=50265):
    """Pre-tokenised per-sample records in the layout of vacnic_amd.data (what the reference's Dataset.__getitem__ produces,
    DSG:524-659, minus the strings): variable-length article / caption ids, fixed-length names_art_ids / names_ids_flatten
    (make_new_entity_ids pads them, DSG:352-354), names_ids rows padded per sample with the <NONAME> row last (DSG:356-358),
    0..max_faces face embeddings, one uint8 image.  Deterministic in (seed, index)."""
    out = []
    for i in range(n):
        g = np.random.default_rng([seed, i])
        la, lc = int(g.integers(8, max_article + 1)), int(g.integers(4, max_caption + 1))
        art = np.concatenate([[bos], g.integers(3, vocab, la - 2), [eos]]).astype(np.int64)
        cap = np.concatenate([[bos], g.integers(3, vocab, lc - 2), [eos]]).astype(np.int64)
        nn_ = int(g.integers(0, max_names + 1))
        rows = [[bos] + g.integers(3, vocab, int(g.integers(1, 5))).tolist() + [eos] for _ in range(nn_)] + [[bos, noname, eos]]
        w = max(len(r) for r in rows)
        names_ids = np.array([r + [pad] * (w - len(r)) for r in rows], dtype=np.int64)
        flat = [bos]
        for r in rows[:-1]:
            flat += r[1:-1] + [ent]
        if nn_ == 0:
            flat += [noname]
        flat = (flat[:gt_len - 1] + [eos])
        flat = np.array(flat + [pad] * (gt_len - len(flat)), dtype=np.int64)
        na = [bos] + g.integers(3, vocab, int(g.integers(0, ner_len - 2))).tolist() + [eos]
        names_art = np.array(na + [pad] * (ner_len - len(na)), dtype=np.int64)
        nf = int(g.integers(0, max_faces + 1))
        faces = (g.standard_normal((nf, 512)) / np.sqrt(512.0)).astype(np.float16)
        img = g.integers(0, 256, (3, image_size, image_size), dtype=np.uint8)
        out.append({"article_ids": art, "caption_ids": cap, "names_art_ids": names_art, "names_ids": names_ids,
                    "names_ids_flatten": flat, "face_emb": faces, "image": img})
    return out
