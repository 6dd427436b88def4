"""Test infrastructure (NOT product code): the weights of the configs[4] parity fixture — batch 1, beam 5, max_length 50,
length_penalty 2.0 (TRAIN:513-520, DDPINF:758-842, run_full_train.sh:10-11).

Why the weights are constructed.  With N(0, 0.02) weights the next-token distribution of a BART decoder is a flat random
field: top-2 margins are far below bf16 resolution, and over 50 positions x 5 beams x 10 candidates a bf16 implementation and
the fp32 reference part ways at some near-tie with near certainty (measured here with a bf16-rounding emulation of the oracle:
0 of 30 seeded inputs kept their ids; first difference at position 2..41).  "Identical ids" is then a coin toss, not a parity
bar.  A trained captioner is peaked instead.  This fixture plants that property: along the caption the model is to emit, the
tied embedding row of the next token gets a component along the decoder's final hidden state at that position (computed by the
oracle, teacher forced; minus the component all states share, which would make every planted row answer at every position), so the intended token wins every position by tens of logit units while everything else in the model
stays the seeded random network (a few decoder matrices rescaled, below): encoder, cross-attention, KV cache, beam reorder,
n-gram bans, forced BOS/EOS, min_length, hypothesis bookkeeping and the length penalty all run as in production, and a fault
in any of them moves the hidden state off the planted direction and changes the ids.

Planted structure (one source, two decoding paths because `forced_bos_token_id=0` changes the prefix):
  * path P (library defaults):     2, p1 .. p48          * path H (hub defaults): 2, 0, q2 .. q48
  * on path P at cur_len T_EOS the EOS row (token 2) is the best candidate and the chain's next token the second best:
    without `min_length` the caption ends there (the other beams run on to max_length and lose by the length-normalised
    score), with `min_length=49` EOS is suppressed and the chain runs to the forced EOS at position 49 — all 50 positions are
    decoded.  Path H has no early EOS (its two cases differ by `min_length` only and decode 50 positions).
  * every planted row is sized so that its token beats the best other token (noise, other planted rows, the previous token's
    self-similarity, EOS) by MARGIN logit units at its position.
The EOS row is also the decoder start token's input embedding, so the construction is iterated to a fixed point.

    python oracle/cfg5_fixture.py        # writes tests/golden/cfg5_planted.npz (ids + rows: INPUT data of the fixture)
    python oracle/cfg5_fixture.py m4     # writes tests/golden/cfg5_planted_m4.npz: the SENSITIVE variant (below)

Variant "m4" (round 4: is the fixture able to catch a fault?).  The 40-unit margin above forgives any fault that moves a logit by
less than ~40 units.  In m4 every planted token wins by only 4 .. 7 logit units (bf16 noise of the decoder stack: < 0.5, still 8x
smaller; the margin varies with the position so that the runner-up beams of neighbouring positions are not tied), EOS beats the
chain token at T_EOS by 1.5, and the hub-defaults path carries an N-GRAM TRAP: the bigram (a, b) of positions 10-11 returns at
positions 30-31, and at position 32 the token c that followed it the first time is planted as the BEST candidate, 2 units above
the chain's real next token — only the no-repeat-3-gram ban keeps the caption on the chain.  tests/test_model_gpu.py mutates the
decoder (a corrupted KV-cache row, a skipped beam reorder, a dropped ban) and asserts that the ids / n-best lists CHANGE.

oracle/make_golden.py::run_generate_cfg5_case then runs transformers' beam search over the REAL reference model with these
weights and stores the sequences (tests/golden/generate_cfg5.npz).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from vacnic_amd import synthetic
from vacnic_amd.config import VacnicConfig

PLANTED = os.path.join(ROOT, "tests", "golden", "cfg5_planted.npz")
PLANTED_M4 = os.path.join(ROOT, "tests", "golden", "cfg5_planted_m4.npz")
# the sensitive variant: margins, the n-gram trap of path H (positions of the first / second occurrence of the bigram)
M4_MARGIN, M4_EOS_OVER, M4_TRAP_OVER = 4.0, 1.5, 2.0
TRAP_I, TRAP_J = 10, 30
# ... and a DECOY on path P: at position DECOY_T a token that occurs nowhere else beats the chain's token by DECOY_OVER, so for one
# step the chain is beam 1; the decoy's continuation is a flat field, the chain's is planted, and at the next step the chain is
# beam 0 again — a beam SWAP (reorder indices [1, 0, ...]): the KV cache of the best hypothesis has to move between rows.  Without
# it every runner-up of this fixture branches off beam 0 and a decoder that never reordered its caches would go unnoticed.
DECOY_T, M4_DECOY_OVER = 36, 2.0


def m4_margin(t):
    """per-position margin of variant m4: 4 .. 7 logit units, never equal at neighbouring positions."""
    return M4_MARGIN + 0.75 * ((7 * t) % 5)

MAX_LENGTH, NUM_BEAMS, LENGTH_PENALTY = 50, 5, 2.0
T_EOS = 24                             # cur_len at which the planted EOS wins on path P
MARGIN = 40.0                          # logit units by which a planted token beats the best other token (bf16 noise here: < 0.5)
EOS_OVER = 15.0                        # ... and by which EOS beats the chain's next token at T_EOS: smaller than MARGIN, so that with
                                       # min_length the chain (one 15-unit loss) still beats every beam that ever left a chain (>= 40)
# Decoder weight scales, chosen by measurement on this model (tools: the statistics printed by `python oracle/cfg5_fixture.py`):
# with plain N(0, 0.02) weights the final hidden state of a position is 79 % parallel to the embedding of the token just fed
# (the model repeats its input, and rows planted along such states would all be parallel).  With these scales the state is a
# random non-linear function of its inputs (self-similarity 0.06), whose token-specific part is 76 % of its norm and depends on
# the decoding HISTORY (same token after another prefix: 37 % relative change) and on the SOURCE (another article / image:
# 56 %); the token-specific parts of different positions are nearly orthogonal (cosine <= 0.23).  A fault in the KV cache, the
# beam reorder or the cross-attention therefore moves the state by tens of percent.
FC2_SCALE, CROSS_SCALE, SELF_SCALE = 8.0, 1.5, 1.5          # decoder fc2 / cross-attention v,out / self-attention v,out
SEED = 9

# (name, generate kwargs) — configs[4]'s call with the library defaults, with the hub checkpoints' generation defaults
# (config.HUB_GENERATION_DEFAULTS), and each with min_length 49 (every one of the 50 positions decoded)
CASES = [("plain", {}),
         ("hub", dict(no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0)),
         ("full50", dict(min_length=49)),
         ("hub_full50", dict(min_length=49, no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0))]


def cfg5_cfg():
    """BART-large width (d=1024, 16 heads, ffn 4096) at 2+2 layers: the persistent decoder-step kernel's slot variant runs
    with the workgroup counts of the real model; small enough for a cache-less CPU beam search."""
    return VacnicConfig(d_model=1024, encoder_layers=2, decoder_layers=2, encoder_attention_heads=16, decoder_attention_heads=16,
                        encoder_ffn_dim=4096, decoder_ffn_dim=4096, enc_fusion_layer=[0, 1], dim_common=1024, clip_width=1024,
                        dropout=0.0)


def inputs(cfg, seed=SEED):
    batch = synthetic.make_batch(cfg, 1, S=40, T=8, F=2, seed=seed, image_size=32)
    img = synthetic._normal("img_cls", (1, cfg.clip_width), 1.0, seed)
    return batch, img


def base_state_dict(cfg):
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    sd["model.shared.weight"] = sd["model.shared.weight"] * synthetic.GEN_SHARPEN
    for k in list(sd):
        if not k.startswith("model.decoder.layers"):
            continue
        if k.endswith("fc2.weight"):
            sd[k] = sd[k] * FC2_SCALE
        elif k.endswith(("encoder_attn.out_proj.weight", "encoder_attn.v_proj.weight")):
            sd[k] = sd[k] * CROSS_SCALE
        elif k.endswith(("self_attn.out_proj.weight", "self_attn.v_proj.weight")):
            sd[k] = sd[k] * SELF_SCALE
    return sd


def state_dict(cfg, planted=None):
    """the fixture's weights: seeded random network + the planted embedding rows."""
    sd = base_state_dict(cfg)
    if planted is None:
        planted = np.load(PLANTED)
    E = sd["model.shared.weight"].clone()
    E[torch.from_numpy(np.asarray(planted["ids"], dtype=np.int64))] = torch.from_numpy(np.asarray(planted["rows"], dtype=np.float32))
    sd["model.shared.weight"] = E
    return sd


def expected(planted=None):
    """the sequences the planted structure is built to produce, per case name."""
    if planted is None:
        planted = np.load(PLANTED)
    P, Hh = planted["path_P"].tolist(), planted["path_H"].tolist()
    if "plain_ends_early" in planted and not bool(planted["plain_ends_early"]):
        # variant m4: with 4-unit margins a position's log-probability is not ~0 (thousands of noise tokens share the softmax), so
        # under length_penalty 2.0 the full-length chain out-scores the hypothesis that ended at T_EOS: the plain case runs on
        return {"plain": P[:MAX_LENGTH - 1] + [2], "full50": P[:MAX_LENGTH - 1] + [2],
                "hub": Hh[:MAX_LENGTH - 1] + [2], "hub_full50": Hh[:MAX_LENGTH - 1] + [2]}
    return {"plain": P[:T_EOS] + [2], "full50": P[:MAX_LENGTH - 1] + [2],
            "hub": Hh[:MAX_LENGTH - 1] + [2], "hub_full50": Hh[:MAX_LENGTH - 1] + [2]}


def plant_m4():
    """variant m4 (see the module docstring): small per-position margins, EOS over the chain by 1.5, n-gram trap on path H.
    Rows are built INCREMENTALLY (a token that occurs at several positions — the trap's bigram, the trap token — gets one component
    per position; the position states are nearly orthogonal) and iterated to a fixed point."""
    from oracle import vacnic_oracle as O
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = cfg5_cfg()
    sd = base_state_dict(cfg)
    E0 = sd["model.shared.weight"]
    E = E0.clone()
    batch, img = inputs(cfg)
    src = batch["article_ids"]
    mask = O.create_src_mask_bart(src)
    kw = dict(name_ids=batch["names_art_ids"], name_mask=O.create_src_mask_bart(batch["names_art_ids"]),
              face_features=batch["face_emb"], face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]))
    g = np.random.default_rng(2024)
    pool = [int(t) for t in g.permutation(np.arange(1000, 50000))[:256]]
    chains = {"P": [2] + pool[:MAX_LENGTH - 2], "H": [2, 0] + pool[64:64 + MAX_LENGTH - 3]}
    Hc = chains["H"]
    Hc[TRAP_J], Hc[TRAP_J + 1] = Hc[TRAP_I], Hc[TRAP_I + 1]                       # the bigram (a, b) returns
    trap_tok, trap_t = Hc[TRAP_I + 2], TRAP_J + 2                                   # c: banned at position TRAP_J + 2 by no_repeat_ngram_size 3
    assert len(set(Hc[2:])) == len(Hc[2:]) - 2 and Hc[trap_t] != trap_tok
    decoy_tok = pool[200]
    assert decoy_tok not in chains["P"] and decoy_tok not in Hc
    ids = sorted(set(chains["P"][1:] + Hc[2:]) | {2, decoy_tok})
    NEG = -1e30

    decoy_over = [None]

    def sweep(write):
        sd["model.shared.weight"] = E
        enc_h = O.encoder(sd, cfg, src, mask, img, kw["name_ids"], kw["name_mask"], kw["face_features"], kw["face_mask"])[0]
        worst, eos_over, trap_over, eos_row = 1e9, None, None, None
        for name, seq in chains.items():
            for t in range(2 if name == "H" else 1, MAX_LENGTH - 1):
                h = O.decoder(sd, cfg, torch.tensor([seq[:t]]), enc_h, mask)[-1][0, -1]
                logits = E @ h
                d = h / h.norm() - mu
                d = d / d.norm()
                c = float(d @ h)
                want = seq[t]
                eos_step = name == "P" and t == T_EOS
                trap_step = name == "H" and t == trap_t
                decoy_step = name == "P" and t == DECOY_T
                others = logits.clone()
                others[want] = NEG
                if decoy_step:
                    others[decoy_tok] = NEG
                    decoy_over[0] = float(logits[decoy_tok] - logits[want])
                if eos_step:
                    others[2] = NEG
                    eos_over = float(logits[2] - logits[want])
                if trap_step:
                    others[trap_tok] = NEG
                    trap_over = float(logits[trap_tok] - logits[want])
                worst = min(worst, float(logits[want] - others.max()) - (m4_margin(t) - M4_MARGIN))
                if write:
                    target = float(others.max()) + m4_margin(t)
                    E[want] = E[want] + d * ((target - float(logits[want])) / c)
                    if eos_step:
                        eos_row = E[2] + d * ((target + M4_EOS_OVER - float(logits[2])) / c)
                    if trap_step:
                        E[trap_tok] = E[trap_tok] + d * ((target + M4_TRAP_OVER - float(logits[trap_tok])) / c)
                    if decoy_step:
                        E[decoy_tok] = E[decoy_tok] + d * ((target + M4_DECOY_OVER - float(logits[decoy_tok])) / c)
        return worst, eos_over, trap_over, eos_row

    with torch.no_grad():
        sd["model.shared.weight"] = E
        enc_h = O.encoder(sd, cfg, src, mask, img, kw["name_ids"], kw["name_mask"], kw["face_features"], kw["face_mask"])[0]
        st0 = O.decoder(sd, cfg, torch.tensor([chains["P"]]), enc_h, mask)[-1][0]
        mu = (st0 / st0.norm(dim=1, keepdim=True)).mean(0)
        for rnd in range(40):
            _, _, _, eos_row = sweep(True)
            E[2] = eos_row
            worst, eos_over, trap_over, _ = sweep(False)
            print(f"round {rnd}: smallest chain margin (minus its position's spread) {worst:.3f}, EOS over the chain token {eos_over:.3f}, "
                  f"trap token over the chain token {trap_over:.3f}, decoy over the chain token {decoy_over[0]:.3f}", flush=True)
            if (worst >= M4_MARGIN - 0.1 and abs(eos_over - M4_EOS_OVER) <= 0.1 and abs(trap_over - M4_TRAP_OVER) <= 0.1
                    and abs(decoy_over[0] - M4_DECOY_OVER) <= 0.1):
                break
        assert worst >= M4_MARGIN - 0.25 and abs(eos_over - M4_EOS_OVER) <= 0.25 and abs(trap_over - M4_TRAP_OVER) <= 0.25, (worst, eos_over, trap_over)
        assert abs(decoy_over[0] - M4_DECOY_OVER) <= 0.25, decoy_over
    planted = {"ids": np.array(ids, dtype=np.int64), "rows": E[torch.tensor(ids)].numpy().astype(np.float32),
               "path_P": np.array(chains["P"], dtype=np.int64), "path_H": np.array(chains["H"], dtype=np.int64),
               "trap": np.array([TRAP_I, TRAP_J, trap_t, trap_tok], dtype=np.int64),
               "decoy": np.array([DECOY_T, decoy_tok], dtype=np.int64)}
    sdf = state_dict(cfg, planted)
    out = O.beam_search_decode(sdf, cfg, src, mask, img, NUM_BEAMS, MAX_LENGTH, LENGTH_PENALTY, forced_eos_token_id=2, **kw)
    planted["plain_ends_early"] = np.array(out.shape[1] == T_EOS + 1)
    exp = expected(planted)
    for name, extra in CASES:
        out, nbest = O.beam_search_decode(sdf, cfg, src, mask, img, NUM_BEAMS, MAX_LENGTH, LENGTH_PENALTY, forced_eos_token_id=2,
                                          return_nbest=True, **extra, **kw)
        assert out[0].tolist() == exp[name], (name, out[0].tolist(), exp[name])
        if name == "plain":                                  # the hypothesis that ended at T_EOS is in the n-best list either way
            assert any(sq == chains["P"][:T_EOS] for _, sq in nbest[0]), [len(sq) for _, sq in nbest[0]]
        print(f"oracle beam search, case {name}: {out.shape[1]} tokens ok; n-best scores {[round(s, 4) for s, _ in nbest[0]]}", flush=True)
    # the trap works: without the ban the hub path leaves the chain at the trap position
    out = O.beam_search_decode(sdf, cfg, src, mask, img, NUM_BEAMS, MAX_LENGTH, LENGTH_PENALTY, forced_eos_token_id=2,
                               early_stopping=True, forced_bos_token_id=0, **kw)
    assert out[0].tolist() != exp["hub"] and int(out[0, trap_t]) == trap_tok, ("the trap must spring without the ban", out[0].tolist())
    print("without no_repeat_ngram_size the caption takes the trap token at position", trap_t, flush=True)
    np.savez_compressed(PLANTED_M4, **planted)
    print("wrote", PLANTED_M4, os.path.getsize(PLANTED_M4), "bytes")


def _plant():
    from oracle import vacnic_oracle as O
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = cfg5_cfg()
    sd = base_state_dict(cfg)
    E0 = sd["model.shared.weight"]
    E = E0.clone()
    batch, img = inputs(cfg)
    src = batch["article_ids"]
    mask = O.create_src_mask_bart(src)
    kw = dict(name_ids=batch["names_art_ids"], name_mask=O.create_src_mask_bart(batch["names_art_ids"]),
              face_features=batch["face_emb"], face_mask=O.create_src_mask_bart(batch["face_emb"][:, :, -1]))
    g = np.random.default_rng(2024)
    pool = [int(t) for t in g.permutation(np.arange(1000, 50000))[:256]]            # token ids of the chains (distinct)
    chains = {"P": [2] + pool[:MAX_LENGTH - 2], "H": [2, 0] + pool[64:64 + MAX_LENGTH - 3]}
    assert len(chains["P"]) == MAX_LENGTH - 1 and len(chains["H"]) == MAX_LENGTH - 1
    ids = sorted(set(chains["P"][1:] + chains["H"][2:]) | {2})
    NEG = -1e30

    def sweep(write):
        """one left-to-right pass over both chains with the current E.  write=True: size every planted row so that its token
        beats the best other token by MARGIN at its position (the rows of later chain tokens do not influence a state, so the
        pass is exact for everything but the shared EOS row, which is the start token's embedding: hence the rounds).
        Returns (smallest margin of a chain token over everything else, margin of EOS over the chain token at T_EOS, new EOS row)."""
        sd["model.shared.weight"] = E
        enc_h = O.encoder(sd, cfg, src, mask, img, kw["name_ids"], kw["name_mask"], kw["face_features"], kw["face_mask"])[0]
        worst, worst_eos, eos_row = 1e9, 1e9, None
        for name, seq in chains.items():
            for t in range(2 if name == "H" else 1, MAX_LENGTH - 1):               # H: position 1 is the forced BOS
                h = O.decoder(sd, cfg, torch.tensor([seq[:t]]), enc_h, mask)[-1][0, -1]
                logits = E @ h
                d = h / h.norm() - mu
                d = d / d.norm()
                c = float(d @ h)
                want = seq[t]
                eos_step = name == "P" and t == T_EOS
                others = logits.clone()
                others[want] = NEG
                if eos_step:
                    others[2] = NEG
                    worst_eos = float(logits[2] - logits[want])
                worst = min(worst, float(logits[want] - others.max()))
                if write:
                    # row = its seeded random row + a component along the state.  (Rows made of the state direction alone turn the
                    # chain into the iteration x -> F(x) of the decoder, which contracts to a fixed point: consecutive states, and
                    # with them all planted rows, become parallel.  As INPUT embeddings the rows must stay mostly random.)
                    E[want] = E0[want] + d * ((float(others.max()) + MARGIN - float(E0[want] @ h)) / c)
                    if eos_step:
                        eos_row = E0[2] + d * ((float(others.max()) + MARGIN + EOS_OVER - float(E0[2] @ h)) / c)
        return worst, worst_eos, eos_row

    with torch.no_grad():
        # the direction every final state shares (LayerNorm bias, mean attention output ...): a fixed vector of the construction
        sd["model.shared.weight"] = E
        enc_h = O.encoder(sd, cfg, src, mask, img, kw["name_ids"], kw["name_mask"], kw["face_features"], kw["face_mask"])[0]
        st0 = O.decoder(sd, cfg, torch.tensor([chains["P"]]), enc_h, mask)[-1][0]
        mu = (st0 / st0.norm(dim=1, keepdim=True)).mean(0)
        for rnd in range(16):
            _, _, eos_row = sweep(True)
            delta = float((eos_row - E[2]).norm() / eos_row.norm())
            E[2] = eos_row                                                           # EOS row = the decoder start token's embedding
            worst, worst_eos, _ = sweep(False)
            print(f"round {rnd}: EOS row moved by {delta:.4f} (relative); smallest chain margin {worst:.2f}, EOS over the chain token {worst_eos:.2f} logit units", flush=True)
            if worst >= MARGIN - 3.0 and worst_eos >= EOS_OVER - 2.0:
                break
        assert worst >= MARGIN - 3.0 and EOS_OVER - 2.0 <= worst_eos <= EOS_OVER + 2.0, (worst, worst_eos)
    planted = {"ids": np.array(ids, dtype=np.int64), "rows": E[torch.tensor(ids)].numpy().astype(np.float32),
               "path_P": np.array(chains["P"], dtype=np.int64), "path_H": np.array(chains["H"], dtype=np.int64)}
    # ---- and the oracle's beam search produces the intended captions
    exp = expected(planted)
    sdf = state_dict(cfg, planted)
    for name, extra in CASES:
        out = O.beam_search_decode(sdf, cfg, src, mask, img, NUM_BEAMS, MAX_LENGTH, LENGTH_PENALTY, forced_eos_token_id=2, **extra, **kw)
        assert out[0].tolist() == exp[name], (name, out[0].tolist(), exp[name])
        print(f"oracle beam search, case {name}: {out.shape[1]} tokens ok", flush=True)
    np.savez_compressed(PLANTED, **planted)
    print("wrote", PLANTED, os.path.getsize(PLANTED), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "m4":
        plant_m4()
    else:
        _plant()
