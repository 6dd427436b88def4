"""CPU (not gpu): the oracle restatement against the golden vectors produced by the REAL reference
(oracle/make_golden.py).  fp32 vs fp32, tolerance 1e-4 relative on slices, 1e-5 on scalars."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vacnic_oracle as O          # noqa: E402
from vacnic_amd import synthetic               # noqa: E402
from vacnic_amd.config import ClipVisionConfig, VacnicConfig   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def small_cfg(**kw):
    base = dict(d_model=768, encoder_layers=2, decoder_layers=2, encoder_attention_heads=12, decoder_attention_heads=12,
                encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=768, dropout=0.0)
    base.update(kw)
    return VacnicConfig(**base)


CASES = {
    "mfull_d768": (small_cfg(), dict(B=3, S=48, T=12, F=3)),
    "mfull_d1024": (small_cfg(d_model=1024, encoder_layers=1, decoder_layers=1, encoder_attention_heads=16,
                              decoder_attention_heads=16, encoder_ffn_dim=2048, decoder_ffn_dim=2048, dim_common=1024,
                              clip_width=1024), dict(B=2, S=40, T=10, F=2)),
    # --prompt_mlp_type mlp --map_size 12 32 16 8 (MFULL:76-108): token-mixing prompt MLP over patch tokens, then visual_map
    "mfull_mlp_d1024": (small_cfg(d_model=1024, encoder_layers=1, decoder_layers=1, encoder_attention_heads=16,
                                  decoder_attention_heads=16, encoder_ffn_dim=2048, decoder_ffn_dim=2048, dim_common=1024,
                                  clip_width=768, prompt_mlp_type="mlp", map_size=[12, 32, 16, 8]), dict(B=2, S=40, T=10, F=2)),
    # --init_attn_weight True (MFULL:1858-1870): three attentions per encoder layer share their weight Parameters
    "mfull_tied_d768": (small_cfg(encoder_layers=2, decoder_layers=1, enc_fusion_layer=[0, 1], init_attn_weight=True),
                        dict(B=2, S=40, T=10, F=2)),
}


def slices(t, n=4096):
    f = t.detach().reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy(), np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()])


def check(name, t, gold, rtol=2e-4, atol=2e-5):
    s, c = slices(t)
    np.testing.assert_allclose(s, gold[name + "_s"], rtol=rtol, atol=atol, err_msg=name)
    np.testing.assert_allclose(c[1:], gold[name + "_c"][1:], rtol=1e-4, err_msg=name + " checksum")


def full_case_inputs(cfg, B, S, T, F):
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    if cfg.init_attn_weight:
        synthetic.apply_init_attn_weight(sd, cfg)
    sd_g = synthetic.make_state_dict(synthetic.guide_bart_param_shapes(cfg), seed=2)
    batch = synthetic.make_batch(cfg, B, S=S, T=T, F=F, seed=7, image_size=32)
    img_cls = synthetic.image_features(cfg, B)
    return sd, sd_g, batch, img_cls


def oracle_losses(sd, sd_g, cfg, batch, img_cls):
    src, tgt = batch["article_ids"], batch["caption_ids"]
    tgt_in = O.shift_tokens_right(tgt, 1, 2)
    src_mask = O.create_src_mask_bart(src)
    out = O.mmbart_forward(sd, cfg, src, src_mask, tgt_in, img_cls, face_features=batch["face_emb"],
                           face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]), name_ids=batch["names_art_ids"],
                           name_mask=O.create_src_mask_bart(batch["names_art_ids"]))
    logits = out["logits"]
    txt = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1), ignore_index=1)
    with torch.no_grad():
        gh = O.guide_bart_forward(sd_g, cfg, src, src_mask, tgt_in)
    colam = O.colam_loss(out["decoder_hidden_states"][-1], gh, tgt, 1.0)
    names = O.get_embedding_ner(sd, cfg, batch["names_ids"])
    secla = O.secla_loss(out["hidden_states_face"], names)
    return out, gh, names, txt, colam, secla, txt + secla + 0.5 * colam


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_full_model(case):
    cfg, dims = CASES[case]
    gold = np.load(os.path.join(G, case + ".npz"))
    sd, sd_g, batch, img_cls = full_case_inputs(cfg, **dims)
    for v in sd.values():
        v.requires_grad_(True)
    out, gh, names, txt, colam, secla, loss = oracle_losses(sd, sd_g, cfg, batch, img_cls)
    for k, v, g in (("txt", txt, gold["txt"]), ("colam", colam, gold["colam"]), ("secla", secla, gold["secla"]), ("loss", loss, gold["loss"])):
        assert abs(v.item() - float(g)) <= 2e-5 * max(1.0, abs(float(g))), (k, v.item(), float(g))
    for key in ("logits", "hidden_states_face", "hidden_states_ner", "hidden_states_img", "encoder_last_hidden_state"):
        check(key, out[key], gold)
    check("dec_last", out["decoder_hidden_states"][-1], gold)
    check("guide_last", gh, gold)
    check("names", names, gold)
    assert np.array_equal(out["logits"].argmax(-1).numpy(), gold["argmax"]), "greedy (teacher-forced) ids must be bit-exact"
    np.testing.assert_allclose(torch.logsumexp(out["logits"], -1).detach().numpy(), gold["lse"], rtol=1e-5)
    loss.backward()
    n = 0
    for key in gold.files:
        if key.startswith("grad:") and key.endswith("_s"):
            pname = key[5:-2]
            check("grad:" + pname, sd[pname].grad, gold, rtol=2e-3, atol=1e-6)
            n += 1
    assert n >= 14


def test_oracle_matches_reference_mvis():
    cfg = small_cfg(only_image=True, enc_fusion_layer=[0, 1])
    gold = np.load(os.path.join(G, "mvis_d768.npz"))
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    for v in sd.values():
        v.requires_grad_(True)
    batch = synthetic.make_batch(cfg, 2, S=32, T=8, seed=8, image_size=32)
    src, tgt = batch["article_ids"], batch["caption_ids"]
    img_cls = synthetic._normal("img_cls", (2, 768), 1.0, 3)
    out = O.mmbart_forward(sd, cfg, src, O.create_src_mask_bart(src), O.shift_tokens_right(tgt, 1, 2), img_cls)
    txt = torch.nn.functional.cross_entropy(out["logits"].reshape(-1, cfg.vocab_size), tgt.reshape(-1), ignore_index=1)
    assert abs(txt.item() - float(gold["txt"])) < 2e-5 * float(gold["txt"])
    check("logits", out["logits"], gold); check("hidden_states_img", out["hidden_states_img"], gold)
    assert np.array_equal(out["logits"].argmax(-1).numpy(), gold["argmax"])
    txt.backward()
    for key in gold.files:
        if key.startswith("grad:") and key.endswith("_s"):
            check("grad:" + key[5:-2], sd[key[5:-2]].grad, gold, rtol=2e-3, atol=1e-6)


def test_oracle_trainer_helpers():
    g = np.load(os.path.join(G, "trainer_helpers.npz"))
    ids = torch.tensor([[0, 5, 6, 2, 1], [0, 9, 2, 1, 1], [0, 7, -100, 2, 1]])
    assert np.array_equal(O.shift_tokens_right(ids, 1, 2).numpy(), g["shift"])
    assert np.array_equal(O.create_src_mask_bart(ids).numpy(), g["mask"])
    np.testing.assert_allclose(O.pool(torch.from_numpy(g["pool_in"]), torch.from_numpy(g["pool_mask"])).numpy(), g["pool_out"], rtol=1e-6)
    assert abs(O.secla_loss(torch.from_numpy(g["secla_face"]), torch.from_numpy(g["secla_ner"])).item() - float(g["secla_out"])) < 1e-5
    x = torch.from_numpy(g["hinge_in"])
    for mg, ref in zip((1.0, 0.3), g["hinge_out"]):
        assert abs(torch.clamp(mg - x, min=0).mean().item() - ref) < 1e-6      # HingeEmbeddingLoss(y=-1) == mean relu(margin - x)
    p = torch.from_numpy(g["adam_p0"]).clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for i in range(6):
        lr = 3e-5 * O.linear_schedule_lambda(i, 2.0, 40.0)
        assert abs(lr - g["adam_lrs"][i]) < 1e-12
        p, m, v = O.adamw_step(p, torch.from_numpy(g["adam_g"][i]), m, v, i + 1, lr)
    np.testing.assert_allclose(p.numpy(), g["adam_p6"], rtol=1e-6, atol=1e-7)


def test_oracle_clip_vit_matches_hf_clip_vision():
    g = np.load(os.path.join(G, "clip_vit_hf.npz"))
    v = ClipVisionConfig(width=128, layers=2, patch_size=16, image_size=64, output_dim=64)
    sd = synthetic.make_state_dict(synthetic.clip_visual_param_shapes(v), seed=4, std=0.05)
    img = synthetic._normal("clip_img", (2, 3, 64, 64), 1.0, 5)
    x, x_cls = O.clip_vit_features(sd, v, img)
    np.testing.assert_allclose(x_cls.numpy(), g["x_cls"], rtol=1e-4, atol=1e-5)


def test_oracle_greedy_decode_consistent_with_teacher_forcing():
    cfg = small_cfg(only_image=True, enc_fusion_layer=[0], encoder_layers=1, decoder_layers=1)
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    batch = synthetic.make_batch(cfg, 2, S=16, T=8, seed=3, image_size=32)
    src = batch["article_ids"]; mask = O.create_src_mask_bart(src)
    img_cls = synthetic._normal("img_cls", (2, 768), 1.0, 3)
    ids = O.greedy_decode(sd, cfg, src, mask, img_cls, max_length=6)
    out = O.mmbart_forward(sd, cfg, src, mask, ids[:, :-1], img_cls)
    assert torch.equal(out["logits"].argmax(-1), ids[:, 1:])


GEN_CASES = [(5, 2.0, {}), (5, 1.0, {}), (1, 1.0, dict(min_length=4)), (5, 1.0, dict(min_length=4)),
             (4, 2.0, dict(min_length=3, no_repeat_ngram_size=2)), (3, 0.5, dict(min_length=5, no_repeat_ngram_size=3, early_stopping=True)),
             # the generation defaults of the facebook/bart-* hub checkpoints (config.HUB_GENERATION_DEFAULTS) at config 5's beam / length penalty
             (5, 2.0, dict(no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0))]


def gen_inputs():
    cfg = small_cfg(encoder_layers=1, decoder_layers=1)
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    sd["model.shared.weight"] = sd["model.shared.weight"] * synthetic.GEN_SHARPEN      # as oracle/make_golden.py::run_generate_case
    batch = synthetic.make_batch(cfg, 2, S=24, T=8, F=2, seed=9, image_size=32)
    img = synthetic._normal("img_cls", (2, 768), 1.0, 3)
    kw = dict(face_features=batch["face_emb"], face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]),
              name_ids=batch["names_art_ids"], name_mask=O.create_src_mask_bart(batch["names_art_ids"]))
    return cfg, sd, batch, img, kw


def test_oracle_beam_search_matches_hf_generate_over_reference_logits():
    """tests/golden/generate_small.npz = transformers-5.15 GenerationMixin beam search driving the REAL reference model;
    the oracle restates the 4.18 bookkeeping (norm="v5" selects the newer length normalisation used by that run)."""
    g = np.load(os.path.join(G, "generate_small.npz"))
    cfg, sd, batch, img, kw = gen_inputs()
    src = batch["article_ids"]; mask = O.create_src_mask_bart(src)
    for i, (nb, lp, extra) in enumerate(GEN_CASES):
        got = O.beam_search_decode(sd, cfg, src, mask, img, nb, 12, lp, forced_eos_token_id=2, norm="v5", **extra, **kw)
        assert np.array_equal(got.numpy(), g[f"seq{i}"]), (i, got.tolist(), g[f"seq{i}"].tolist())


def test_checkpoint_written_by_the_product_is_read_back_by_the_reference():
    """tests/golden/checkpoint_readback.npz = logits of the REAL reference class after
    `load_state_dict(ck["model"], strict=True)` of a checkpoint written by vacnic_amd/checkpoint.py (oracle/make_golden.py
    ::run_checkpoint_readback).  Here the same checkpoint is rebuilt (seeded) and replayed through the oracle: the tensors the
    product writes under the reference's names are exactly what the reference reads."""
    from oracle import ckpt_case
    g = np.load(os.path.join(G, "checkpoint_readback.npz"))
    cfg, ck, batch, img = ckpt_case.make_checkpoint()
    assert len(ck["model"]) == int(g["n_keys"]) and ck["meta"]["step"] == int(g["step"])
    assert {"model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight", "lm_head.weight", "final_logits_bias"} <= set(ck["model"])
    assert torch.equal(ck["model"]["lm_head.weight"], ck["model"]["model.shared.weight"])
    sd = {k: v for k, v in ck["model"].items() if k in synthetic.mmbart_param_shapes(cfg)}
    src, tgt = batch["article_ids"], batch["caption_ids"]
    with torch.no_grad():
        out = O.mmbart_forward(sd, cfg, src, O.create_src_mask_bart(src), O.shift_tokens_right(tgt, 1, 2), img,
                               face_features=batch["face_emb"], face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]),
                               name_ids=batch["names_art_ids"], name_mask=O.create_src_mask_bart(batch["names_art_ids"]))
    check("logits", out["logits"], g)
    assert np.array_equal(out["logits"].argmax(-1).numpy(), g["argmax"])


def test_oracle_config5_beam_search_matches_reference_golden():
    """configs[4] (batch 1, beam 5, max_length 50, length_penalty 2.0): tests/golden/generate_cfg5.npz = transformers' beam search
    over the REAL reference model with the planted-caption weights of oracle/cfg5_fixture.py; the oracle's restatement of the
    4.18 bookkeeping must give the same ids (two of the four cases here; the GPU test covers all four against the golden)."""
    from oracle import cfg5_fixture as F5
    g = np.load(os.path.join(G, "generate_cfg5.npz"))
    planted = np.load(F5.PLANTED)
    exp = F5.expected(planted)
    cfg = F5.cfg5_cfg()
    sd = F5.state_dict(cfg, planted)
    batch, img = F5.inputs(cfg)
    src = batch["article_ids"]; mask = O.create_src_mask_bart(src)
    kw = dict(face_features=batch["face_emb"], face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]),
              name_ids=batch["names_art_ids"], name_mask=O.create_src_mask_bart(batch["names_art_ids"]))
    for name, extra in F5.CASES:
        assert g[name][0].tolist() == exp[name], name
        if name in ("plain", "hub_full50"):
            out = O.beam_search_decode(sd, cfg, src, mask, img, F5.NUM_BEAMS, F5.MAX_LENGTH, F5.LENGTH_PENALTY, forced_eos_token_id=2,
                                       **extra, **kw)
            assert np.array_equal(out.numpy(), g[name]), (name, out.tolist(), g[name].tolist())
    assert g["plain"].shape[1] == F5.T_EOS + 1 and g["full50"].shape[1] == 50 and g["hub_full50"].shape[1] == 50


def _nbest_sums(seqs, scores, lp):
    """(sum of log-probabilities, ids without padding) per returned hypothesis of a reference n-best golden."""
    out = []
    for sq, sc in zip(seqs, scores):
        ids = [int(t) for t in sq if int(t) != 1]
        out.append((float(sc) * (len(ids) - 1) ** lp, ids))
    return out


def test_oracle_config5_sensitive_fixture_nbest_matches_reference_golden():
    """the SENSITIVE configs[4] fixture (oracle/cfg5_fixture.py variant m4: 4 .. 7-unit margins, EOS 1.5 over the chain, n-gram
    trap): the oracle's beam search must reproduce the reference's best sequence AND its whole n-best list (ids and scores of all
    five final beams, tests/golden/generate_cfg5_m4.npz), and the reference itself walks into the trap without the ban."""
    from oracle import cfg5_fixture as F5
    g = np.load(os.path.join(G, "generate_cfg5_m4.npz"))
    planted = np.load(F5.PLANTED_M4)
    exp = F5.expected(planted)
    trap_t, trap_tok = int(planted["trap"][2]), int(planted["trap"][3])
    assert g["hub_without_ngram_ban"][0].tolist() != exp["hub"] and int(g["hub_without_ngram_ban"][0, trap_t]) == trap_tok
    cfg = F5.cfg5_cfg()
    sd = F5.state_dict(cfg, planted)
    batch, img = F5.inputs(cfg)
    src = batch["article_ids"]; mask = O.create_src_mask_bart(src)
    kw = dict(face_features=batch["face_emb"], face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]),
              name_ids=batch["names_art_ids"], name_mask=O.create_src_mask_bart(batch["names_art_ids"]))
    for name, extra in F5.CASES:
        assert g[name][0].tolist() == exp[name], name
        if name in ("plain", "hub_full50"):
            out, nbest = O.beam_search_decode(sd, cfg, src, mask, img, F5.NUM_BEAMS, F5.MAX_LENGTH, F5.LENGTH_PENALTY, forced_eos_token_id=2,
                                              return_nbest=True, **extra, **kw)
            assert out[0].tolist() == g[name][0].tolist(), name
            want = _nbest_sums(g[name + "_nbest"], g[name + "_nbest_scores"], F5.LENGTH_PENALTY)
            assert len(nbest[0]) == len(want) == F5.NUM_BEAMS
            for (sc, ids), (wsum, wids) in zip(nbest[0], want):          # fp32 against fp32: same order, same ids, same scores
                got_ids = ids + [2] if len(ids) < F5.MAX_LENGTH and wids[-1] == 2 and ids[-1] != 2 else ids
                assert got_ids == wids, (name, got_ids, wids)
                assert abs(sc * len(ids) ** F5.LENGTH_PENALTY - wsum) <= 2e-3 * max(1.0, abs(wsum)), (name, sc * len(ids) ** 2, wsum)
