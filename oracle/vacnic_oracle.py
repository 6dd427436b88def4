"""CPU oracle for the VACNIC training step — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch fp32 restatement (CPU, autograd for gradients) of the reference's algorithm for the
hot path, written as pure functions over a state dict that uses the REFERENCE's parameter names.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product (vacnic_amd/) never does.

Pinning: tests/golden/*.npz hold outputs of the REAL reference modules (imported from
/root/reference in the authoring container by oracle/make_golden.py, which is committed) on seeded
inputs; tests/test_oracle.py checks this restatement against them.  The reference has no tests or
golden vectors of its own.  Third-party arithmetic absent from /root/reference:
openai-CLIP `clip==1.0` VisionTransformer (restated from its published model.py; cross-checked in
make_golden.py against transformers.CLIPVisionModel) and transformers==4.18 BartForConditionalGeneration
(identical arithmetic to the vendored MFULL classes with fusion disabled).

Citations: MFULL = src/models/modeling_mmbart_clip_inside_vis_clipcap_ent_type_final_fix_len_enc_self_face_name_ids_crossattn.py,
MVIS = ..._enc_self_crossattn.py, TRAIN = train_mmbart_enc_self_face_name_ids_retrieve_crossattn_bart_guide_match.py.
"""
import math

import torch
import torch.nn.functional as Fn

FMIN = torch.finfo(torch.float32).min


# ------------------------------------------------------------------------------------------ helpers
def shift_tokens_right(input_ids, pad_token_id, decoder_start_token_id):
    """TRAIN:196-209."""
    out = input_ids.new_zeros(input_ids.shape)
    out[:, 1:] = input_ids[:, :-1].clone()
    out[:, 0] = decoder_start_token_id
    out.masked_fill_(out == -100, pad_token_id)
    return out


def create_src_mask_bart(input_ids):
    """TRAIN:212-217: 1 where the id differs from pad (=1).  Also used on face_emb[:, :, -1] (TRAIN:269)."""
    return (input_ids != 1).to(torch.int64)


def expand_mask(mask, tgt_len=None):
    """_expand_mask, MFULL:387-398: [B,S] 0/1 -> additive [B,1,T,S] with finfo.min at masked keys."""
    B, S = mask.shape
    T = tgt_len if tgt_len is not None else S
    inv = 1.0 - mask[:, None, None, :].expand(B, 1, T, S).to(torch.float32)
    return inv.masked_fill(inv.to(torch.bool), FMIN)


def causal_mask(T):
    """_make_causal_mask, MFULL:373-385."""
    m = torch.full((T, T), FMIN)
    c = torch.arange(T)
    m.masked_fill_(c < (c + 1).view(T, 1), 0)
    return m[None, None]


def linear(sd, prefix, x):
    return Fn.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def layer_norm(sd, prefix, x):
    return Fn.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def attention(sd, prefix, hidden, num_heads, key_value_states=None, attention_mask=None):
    """BartAttention.forward without cache, MFULL:454-565."""
    B, T, d = hidden.shape
    hd = d // num_heads
    q = linear(sd, prefix + ".q_proj", hidden) * hd ** -0.5
    src = hidden if key_value_states is None else key_value_states
    k = linear(sd, prefix + ".k_proj", src)
    v = linear(sd, prefix + ".v_proj", src)

    def shape(t):
        return t.view(B, -1, num_heads, hd).transpose(1, 2)

    w = torch.matmul(shape(q), shape(k).transpose(-1, -2))           # [B,H,T,S]
    if attention_mask is not None:
        w = w + attention_mask
    w = torch.softmax(w, dim=-1)
    o = torch.matmul(w, shape(v)).transpose(1, 2).reshape(B, T, d)
    return linear(sd, prefix + ".out_proj", o)


def ffn_block(sd, up, down, norm, x):
    """residual FFN, post-LN: LN(x + down(gelu(up(x))))  (MFULL:647-653, 658-664, 738-744)."""
    return layer_norm(sd, norm, x + linear(sd, down, Fn.gelu(linear(sd, up, x))))


# ------------------------------------------------------------------------------------------ encoder
def encoder_layer(sd, L, cfg, h, attn_mask, img, face, ner, face_name_mask, img_ner_mask, fused):
    """BartEncoderLayer.forward, MFULL:618-762 (only_image: MVIS:591-690)."""
    H = cfg.encoder_attention_heads
    if fused:
        img = ffn_block(sd, L + "._linear_1up", L + "._linear_1down", L + ".img_layer_norm", img)          # :647-653
        if not cfg.only_image:
            face = ffn_block(sd, L + "._face_up", L + "._face_down", L + ".face_layer_norm", face)         # :658-664
            ner_face = torch.cat((face, ner), dim=1)                                                       # :668
            ner = layer_norm(sd, L + ".img_name_attn_layer_norm",
                             ner + attention(sd, L + ".self_attn_img_name", ner, H, ner_face, face_name_mask))  # :669-679
            B, N, d = ner.shape
            pre = Fn.gelu(linear(sd, L + ".ner_map_up", ner.reshape(B, d, N)))      # flat view, NOT a transpose (:683)
            pre = linear(sd, L + ".ner_map_down", pre).reshape(B, cfg.max_ner_type_len_gt, d)              # :685-687
            pre = layer_norm(sd, L + ".ner_map_layer_norm", pre)                                           # :688
            kv = torch.cat((img, pre), dim=1)                                                              # :691
        else:
            kv = img
        h = layer_norm(sd, L + ".self_attn_layer_norm", h + attention(sd, L + ".self_attn", h, H, None, attn_mask))   # :697-707
        h = layer_norm(sd, L + ".img_ner_attn_layer_norm",
                       h + attention(sd, L + ".cross_attn_img_ner", h, H, kv, img_ner_mask))                          # :711-723
    else:
        h = layer_norm(sd, L + ".self_attn_layer_norm", h + attention(sd, L + ".self_attn", h, H, None, attn_mask))   # :726-736
    h = ffn_block(sd, L + ".fc1", L + ".fc2", L + ".final_layer_norm", h)                                             # :738-744
    return h, img, face, ner


def embed(sd, tok, pos, norm, ids, scale):
    """LN(embed(ids)*scale + pos[arange+2]), MFULL:1243-1249 (offset 2: MFULL:401-418)."""
    T = ids.shape[1]
    x = Fn.embedding(ids, sd[tok], padding_idx=1) * scale + sd[pos][2:2 + T][None]   # nn.Embedding(.., padding_idx): no grad to the pad row
    return layer_norm(sd, norm, x)


def encoder(sd, cfg, input_ids, attention_mask, image_features, name_ids=None, name_mask=None, face_features=None,
            face_mask=None):
    """BartEncoder.forward, MFULL:1172-1381 (add_ner_ffn=True path)."""
    e = "model.encoder"
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    B, S = input_ids.shape
    h = embed(sd, "model.shared.weight", e + ".embed_positions.weight", e + ".layernorm_embedding", input_ids, scale)
    face = ner = face_name_mask = None
    if not cfg.only_image:
        ner = embed(sd, e + ".embed_tokens_ner.weight", e + ".embed_positions_ner.weight", e + ".layernorm_embedding_ner",
                    name_ids, scale)                                                                      # :1254-1260
        fm = torch.cat((face_mask, name_mask), dim=1)                                                     # :1262
        face_name_mask = expand_mask(fm, tgt_len=cfg.max_ner_type_len)                                    # :1264
        face = linear(sd, e + "._linear_1", face_features)                                                # :1269
    if cfg.prompt_mlp_type == "mlp":
        # MLP.forward, MFULL:76-108: the [B, tokens, width] features are RESHAPED (not transposed) to [B, width, tokens], Linear(+Tanh)
        # runs over the last axis, and the result is reshaped back to [B, map_size[-1], width]
        _, feat, hid = image_features.shape
        x = image_features.reshape(B, hid, feat)
        nl = len(cfg.map_size) - 1
        for i in range(nl):
            x = linear(sd, f"{e}.prompt_mlp.model.{2 * i}", x)
            if i < nl - 1:
                x = torch.tanh(x)
        img = x.reshape(B, cfg.map_size[-1], hid)                                                         # :1274
    else:
        img = linear(sd, e + ".prompt_mlp.model.2", torch.tanh(linear(sd, e + ".prompt_mlp.model.0", image_features)))
        img = img.reshape(B, cfg.prompt_size, 768)                                                        # :1274-1276
    if cfg.d_model == 1024:
        img = linear(sd, e + ".visual_map", img)                                                          # :1277-1278
    n_kv = cfg.prompt_len + (0 if cfg.only_image else cfg.max_ner_type_len_gt)
    img_ner_mask = expand_mask(torch.ones(B, n_kv), tgt_len=S)                                            # :1286-1296
    attn_mask = expand_mask(attention_mask)
    for i in range(cfg.encoder_layers):
        h, img, face, ner = encoder_layer(sd, f"{e}.layers.{i}", cfg, h, attn_mask, img, face, ner, face_name_mask,
                                          img_ner_mask, i in cfg.enc_fusion_layer)
    return h, img, ner, face


def decoder(sd, cfg, dec_ids, enc_h, enc_mask, prefix="model.decoder", shared="model.shared.weight"):
    """BartDecoder.forward + BartDecoderLayer.forward, MFULL:1453-1675, 793-890 (no decoder_attention_mask,
    as in TRAIN:281).  Returns all hidden states (output_hidden_states=True, TRAIN:743)."""
    H = cfg.decoder_attention_heads
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    B, T = dec_ids.shape
    h = embed(sd, shared, prefix + ".embed_positions.weight", prefix + ".layernorm_embedding", dec_ids, scale)
    self_mask = causal_mask(T) if T > 1 else None
    cross_mask = expand_mask(enc_mask, tgt_len=T)
    states = [h]
    for i in range(cfg.decoder_layers):
        L = f"{prefix}.layers.{i}"
        h = layer_norm(sd, L + ".self_attn_layer_norm", h + attention(sd, L + ".self_attn", h, H, None, self_mask))
        h = layer_norm(sd, L + ".encoder_attn_layer_norm", h + attention(sd, L + ".encoder_attn", h, H, enc_h, cross_mask))
        h = ffn_block(sd, L + ".fc1", L + ".fc2", L + ".final_layer_norm", h)
        states.append(h)
    return states


def mmbart_forward(sd, cfg, input_ids, attention_mask, decoder_input_ids, image_features, face_features=None,
                   face_mask=None, name_ids=None, name_mask=None):
    """BartForMultiModalGeneration.forward, MFULL:1929-2021 (logits = lm_head(h) + final_logits_bias(=0))."""
    enc_h, img, ner, face = encoder(sd, cfg, input_ids, attention_mask, image_features, name_ids, name_mask, face_features,
                                    face_mask)
    states = decoder(sd, cfg, decoder_input_ids, enc_h, attention_mask)
    logits = Fn.linear(states[-1], sd["model.shared.weight"])            # lm_head tied to shared (MFULL:1885)
    return {"logits": logits, "decoder_hidden_states": states, "encoder_last_hidden_state": enc_h,
            "hidden_states_face": face, "hidden_states_ner": ner, "hidden_states_img": img}


def guide_bart_forward(sd, cfg, input_ids, attention_mask, decoder_input_ids):
    """Frozen HF BartForConditionalGeneration (TRAIN:745-751,293-294): vanilla encoder + decoder;
    only decoder_hidden_states[-1] is consumed, so the LM head is skipped."""
    e = "model.encoder"
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    h = embed(sd, "model.shared.weight", e + ".embed_positions.weight", e + ".layernorm_embedding", input_ids, scale)
    m = expand_mask(attention_mask)
    for i in range(cfg.encoder_layers):
        L = f"{e}.layers.{i}"
        h = layer_norm(sd, L + ".self_attn_layer_norm", h + attention(sd, L + ".self_attn", h, cfg.encoder_attention_heads, None, m))
        h = ffn_block(sd, L + ".fc1", L + ".fc2", L + ".final_layer_norm", h)
    return decoder(sd, cfg, decoder_input_ids, h, attention_mask)[-1]


# ------------------------------------------------------------------------------------------ CLIP ViT
def clip_vit_features(sd, vcfg, img):
    """extract_clip_img_feat, TRAIN:220-240, over openai-CLIP VisionTransformer (clip/model.py):
    returns (ln_post(patches), ln_post(cls)) in fp32, no `proj`."""
    w, Hh = vcfg.width, vcfg.heads
    x = Fn.conv2d(img, sd["conv1.weight"], stride=vcfg.patch_size)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([sd["class_embedding"] + torch.zeros(x.shape[0], 1, w), x], dim=1) + sd["positional_embedding"]
    x = layer_norm(sd, "ln_pre", x)
    B, T, _ = x.shape
    for i in range(vcfg.layers):
        L = f"transformer.resblocks.{i}"
        y = layer_norm(sd, L + ".ln_1", x)
        qkv = Fn.linear(y, sd[L + ".attn.in_proj_weight"], sd[L + ".attn.in_proj_bias"])
        q, k, v = qkv.split(w, dim=-1)
        sh = lambda t: t.view(B, T, Hh, 64).transpose(1, 2)
        a = torch.softmax(torch.matmul(sh(q) * 64 ** -0.5, sh(k).transpose(-1, -2)), dim=-1)
        o = torch.matmul(a, sh(v)).transpose(1, 2).reshape(B, T, w)
        x = x + linear(sd, L + ".attn.out_proj", o)
        y = linear(sd, L + ".mlp.c_fc", layer_norm(sd, L + ".ln_2", x))
        x = x + linear(sd, L + ".mlp.c_proj", y * torch.sigmoid(1.702 * y))          # QuickGELU
    return layer_norm(sd, "ln_post", x[:, 1:, :]), layer_norm(sd, "ln_post", x[:, 0, :])


# -------------------------------------------------------------------------------------------- losses
def pool(last_hidden_states, attention_mask):
    """TRAIN:178-182."""
    lh = last_hidden_states.masked_fill(~attention_mask[..., None].bool(), 0.0)
    emb = lh.sum(dim=1) / attention_mask.sum(dim=1)[..., None]
    return torch.nan_to_num(emb, nan=1.0)


def colam_loss(dec_h, guide_h, tgt_ids, margin):
    """TRAIN:296-307 + HingeEmbeddingLoss(margin) with target -1 (TRAIN:820)."""
    tm = create_src_mask_bart(tgt_ids)
    a = pool(dec_h, tm); b = pool(guide_h, tm)
    a = a / a.norm(dim=1, keepdim=True); b = b / b.norm(dim=1, keepdim=True)
    scores = torch.matmul(a, b.t())
    return Fn.hinge_embedding_loss(scores.diag(), -torch.ones(a.shape[0]), margin=margin)


def batch_softmax(m):
    """TRAIN:631-647."""
    bs, _, ns, _ = m.shape
    logits = m.max(-1).values.sum(-1) / ns
    return Fn.cross_entropy(logits, torch.arange(bs))


def secla_loss(face_j, ner_j):
    """BatchSoftmax.forward, TRAIN:654-660."""
    m1 = torch.matmul(ner_j.unsqueeze(1), face_j.permute(0, 2, 1))
    m2 = torch.matmul(face_j.unsqueeze(1), ner_j.permute(0, 2, 1))
    return batch_softmax(m1) + batch_softmax(m2)


def get_embedding_ner(sd, cfg, ner_ids_3d):
    """TRAIN:112-133: per name, unmasked mean over its tokens of LN(embed_ner*scale + pos); no grad."""
    e = "model.encoder"
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    with torch.no_grad():
        outs = []
        for i in range(ner_ids_3d.shape[1]):
            h = embed(sd, e + ".embed_tokens_ner.weight", e + ".embed_positions_ner.weight", e + ".layernorm_embedding_ner",
                      ner_ids_3d[:, i, :], scale)
            outs.append(h.mean(dim=1))
        return torch.stack(outs, dim=1)


def train_losses(sd, sd_guide, sd_clip, cfg, vcfg, batch, margin=1.0, alpha=0.5, mapping_loss_weight=1.0, use_secla=True):
    """One forward of train_epoch's loss (TRAIN:253-363) in eval-mode arithmetic (dropout off)."""
    src, tgt = batch["article_ids"], batch["caption_ids"]
    tgt_in = shift_tokens_right(tgt, cfg.pad_token_id, cfg.eos_token_id)                     # TRAIN:267
    src_mask = create_src_mask_bart(src)
    with torch.no_grad():
        _, img_cls = clip_vit_features(sd_clip, vcfg, batch["img_tensor"])                   # TRAIN:274-276
    kw = {}
    if not cfg.only_image:
        kw = dict(face_features=batch["face_emb"], face_mask=create_src_mask_bart(batch["face_emb"][:, :, -1]),
                  name_ids=batch["names_art_ids"], name_mask=create_src_mask_bart(batch["names_art_ids"]))
    out = mmbart_forward(sd, cfg, src, src_mask, tgt_in, img_cls, **kw)
    logits = out["logits"]
    txt = Fn.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1), ignore_index=cfg.pad_token_id)   # TRAIN:287
    res = {"txt": txt, "logits": logits, "out": out}
    total = txt
    if sd_guide is not None:
        with torch.no_grad():
            gh = guide_bart_forward(sd_guide, cfg, src, src_mask, tgt_in)                    # TRAIN:293-294
        res["colam"] = colam_loss(out["decoder_hidden_states"][-1], gh, tgt, margin)
        total = total + alpha * res["colam"]
    if use_secla and not cfg.only_image:
        names = get_embedding_ner(sd, cfg, batch["names_ids"])                               # TRAIN:327
        res["secla"] = secla_loss(out["hidden_states_face"], names)                          # TRAIN:329
        total = total + mapping_loss_weight * res["secla"]
    res["loss"] = total                                                                      # TRAIN:363
    return res


# ------------------------------------------------------------------------------------------ optimizer
def linear_schedule_lambda(step, warmup, total):
    """transformers.get_linear_schedule_with_warmup (TRAIN:102)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update (TRAIN:91), step is 1-based."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step; bc2 = 1 - beta2 ** step
    p = p - (lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps)
    return p, m, v


def greedy_decode(sd, cfg, input_ids, attention_mask, image_features, max_length, **kw):
    """Cache-less greedy decoding: stepwise argmax over oracle logits (greedy ids pin)."""
    B = input_ids.shape[0]
    enc_h, _, _, _ = encoder(sd, cfg, input_ids, attention_mask, image_features, kw.get("name_ids"), kw.get("name_mask"),
                             kw.get("face_features"), kw.get("face_mask"))
    ids = torch.full((B, 1), cfg.decoder_start_token_id, dtype=torch.long)
    for _ in range(max_length - 1):
        h = decoder(sd, cfg, ids, enc_h, attention_mask)[-1]
        nxt = Fn.linear(h[:, -1], sd["model.shared.weight"]).argmax(-1)
        ids = torch.cat([ids, nxt[:, None]], dim=1)
    return ids


# ------------------------------------------------------------------------------------ beam search (config 5)
class _BeamHyps:
    """transformers==4.18 BeamHypotheses (generation_beam_search.py): n-best list scored by
    sum_logprobs / len(hyp)**length_penalty, where len counts the decoder start token and NOT the closing EOS.
    norm="v5" divides by (len - 1)**length_penalty instead (newer transformers subtract the 1-token decoder prompt);
    that switch exists only to pin this restatement against the transformers 5.15 run in oracle/make_golden.py."""

    def __init__(self, num_beams, length_penalty, early_stopping, norm="v4.18"):
        self.n, self.lp, self.early, self.norm = num_beams, length_penalty, early_stopping, norm
        self.beams, self.worst = [], 1e9

    def _len(self, L):
        return L if self.norm == "v4.18" else max(L - 1, 1)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (self._len(len(hyp)) ** self.lp)
        if len(self.beams) < self.n or score > self.worst:
            self.beams.append((score, list(hyp)))
            if len(self.beams) > self.n:
                srt = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst = srt[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.beams) < self.n:
            return False
        if self.early:
            return True
        return self.worst >= best_sum_logprobs / self._len(cur_len) ** self.lp


def banned_ngram_tokens(seq, n):
    """NoRepeatNGramLogitsProcessor: tokens that would complete an n-gram already present in seq."""
    if n <= 0 or len(seq) + 1 < n:
        return []
    prefix = tuple(seq[len(seq) - (n - 1):]) if n > 1 else ()
    return [seq[i + n - 1] for i in range(len(seq) - n + 1) if tuple(seq[i:i + n - 1]) == prefix]


def beam_search_bookkeeping(step_logprobs_fn, B, num_beams, max_length, eos, pad, start, length_penalty=1.0,
                            early_stopping=False, no_repeat_ngram_size=0, min_length=0, forced_eos_token_id=None, norm="v4.18",
                            forced_bos_token_id=None, return_nbest=False):
    """GenerationMixin.beam_search + BeamSearchScorer.process/finalize of transformers 4.18, driven by a callback
    `step_logprobs_fn(seqs) -> [B*num_beams, V]` log-softmax of the next-token logits for the current beams."""
    seqs = [[start] for _ in range(B * num_beams)]
    beam_scores = torch.zeros(B, num_beams, dtype=torch.float32)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHyps(num_beams, length_penalty, early_stopping, norm) for _ in range(B)]
    done = [False] * B
    cur_len = 1
    while True:
        lp = step_logprobs_fn(seqs).clone()                              # [B*nb, V] log_softmax
        V = lp.shape[-1]
        for r, s in enumerate(seqs):                                     # logits processors, HF order
            for tok in banned_ngram_tokens(s, no_repeat_ngram_size):
                lp[r, tok] = -float("inf")
            if cur_len < min_length:
                lp[r, eos] = -float("inf")
            if forced_bos_token_id is not None and cur_len == 1:             # ForcedBOSTokenLogitsProcessor
                lp[r] = -float("inf"); lp[r, forced_bos_token_id] = 0.0
            if forced_eos_token_id is not None and cur_len == max_length - 1:
                keep = lp[r, forced_eos_token_id].clone()
                lp[r] = -float("inf"); lp[r, forced_eos_token_id] = 0.0 if keep == keep else 0.0
        scores = (lp + beam_scores[:, None]).view(B, num_beams * V)
        top_s, top_i = torch.topk(scores, 2 * num_beams, dim=1, largest=True, sorted=True)
        next_idx = torch.div(top_i, V, rounding_mode="floor"); next_tok = top_i % V
        new_seqs, new_scores = [], []
        for b in range(B):
            if done[b]:
                new_seqs += [[pad] * 0 + seqs[b * num_beams] + [pad]] * num_beams
                new_scores += [0.0] * num_beams
                continue
            chosen = []
            for rank in range(2 * num_beams):
                tok, sc, bi = int(next_tok[b, rank]), float(top_s[b, rank]), int(next_idx[b, rank])
                src = b * num_beams + bi
                if tok == eos:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(seqs[src], sc)
                else:
                    chosen.append((sc, tok, src))
                if len(chosen) == num_beams:
                    break
            done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
            for sc, tok, src in chosen:
                new_seqs.append(seqs[src] + [tok]); new_scores.append(sc)
        seqs = new_seqs
        beam_scores = torch.tensor(new_scores, dtype=torch.float32)
        cur_len += 1
        if all(done) or cur_len >= max_length:
            break
    out, nbest = [], []
    for b in range(B):                                                   # finalize
        if not done[b]:
            for j in range(num_beams):
                hyps[b].add(seqs[b * num_beams + j], float(beam_scores[b * num_beams + j]))
        ranked = sorted(hyps[b].beams, key=lambda x: x[0])
        out.append(ranked[-1][1])
        nbest.append([(float(sc), list(sq)) for sc, sq in reversed(ranked)])     # BeamSearchScorer.finalize with num_return_sequences = num_beams
    L = min(max(len(o) for o in out) + 1, max_length)
    res = torch.full((B, L), pad, dtype=torch.long)
    for b, o in enumerate(out):
        res[b, :len(o)] = torch.tensor(o)
        if len(o) < L:
            res[b, len(o)] = eos
    return (res, nbest) if return_nbest else res


def beam_search_decode(sd, cfg, input_ids, attention_mask, image_features, num_beams, max_length, length_penalty=1.0, **kw):
    """Cache-less beam search over the oracle model (config 5: beam 5, max_length 50, length_penalty 2.0)."""
    gen = {k: kw.pop(k) for k in ("early_stopping", "no_repeat_ngram_size", "min_length", "forced_eos_token_id", "norm",
                                  "forced_bos_token_id", "return_nbest") if k in kw}
    B = input_ids.shape[0]
    enc_h, _, _, _ = encoder(sd, cfg, input_ids, attention_mask, image_features, kw.get("name_ids"), kw.get("name_mask"),
                             kw.get("face_features"), kw.get("face_mask"))
    enc_b = enc_h.repeat_interleave(num_beams, dim=0)
    mask_b = attention_mask.repeat_interleave(num_beams, dim=0)

    def step(seqs):
        ids = torch.tensor(seqs, dtype=torch.long)
        h = decoder(sd, cfg, ids, enc_b, mask_b)[-1][:, -1]
        return torch.log_softmax(Fn.linear(h, sd["model.shared.weight"]), dim=-1)

    with torch.no_grad():
        return beam_search_bookkeeping(step, B, num_beams, max_length, cfg.eos_token_id, cfg.pad_token_id,
                                       cfg.decoder_start_token_id, length_penalty, **gen)


# ------------------------------------------------------------------------------------------------ input pipeline
def collate_restated(samples, pad=1, bos=0, eos=2, noname=50266):
    """Pure-python restatement of collate_fn_goodnews_entity_type for the fields the step reads (DSG:22-127) and its helpers
    pad_sequence (DSG:201-216), pad_sequence_from_list (DSG:169-190) and pad_tensor_feat (DSG:272-305), sample by sample like the
    reference.  Pinned by tests/golden/collate.npz (the reference's own functions run on the same records)."""
    def pad_sequence(seqs):                                    # DSG:201-216 with max_len = longest in the batch
        m = max(len(s) for s in seqs)
        return torch.tensor([list(s) + [pad] * (m - len(s)) for s in seqs], dtype=torch.long)
    out = {k: pad_sequence([s[k].tolist() for s in samples]) for k in ("article_ids", "caption_ids", "names_art_ids", "names_ids_flatten")}
    lists = [s["names_ids"].tolist() for s in samples]
    max_len = max(len(seq) for sl in lists for seq in sl)      # get_max_len_list, DSG:131-137
    max_num = max(len(sl) for sl in lists)
    rows_all = []
    for sl in lists:                                           # DSG:169-190
        rows = [seq + [pad] * (max_len - len(seq)) for seq in sl]
        rows += [[bos, noname, eos] + [pad] * (max_len - 3)] * (max_num - len(sl))
        rows_all.append(rows)
    out["names_ids"] = torch.tensor(rows_all, dtype=torch.long)
    lens = [s["face_emb"].shape[0] for s in samples]           # DSG:272-305
    m = max(lens)
    faces = []
    for s, n in zip(samples, lens):
        f = torch.from_numpy(s["face_emb"].astype("float32"))
        if m == 0:
            faces.append(torch.ones((1, 512)))
        elif n < m:
            faces.append(torch.cat((f, torch.ones((m - n, 512))), dim=0))
        else:
            faces.append(f)
    out["face_emb"] = torch.stack(faces).float()
    return out
