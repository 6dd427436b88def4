"""MI355X-native multimodal BART of VACNIC — same class names, constructor kwargs, parameter names
(state_dict-compatible) and forward kwargs as the reference's `src/models` operator surface:

  MFULL = src/models/modeling_mmbart_clip_inside_vis_clipcap_ent_type_final_fix_len_enc_self_face_name_ids_crossattn.py
  MVIS  = src/models/modeling_mmbart_clip_inside_vis_clipcap_ent_type_final_fix_len_enc_self_crossattn.py (only_image=True)

but every operation is a hand-written gfx950 kernel (vacnic_amd.ops).  nn.Linear / nn.LayerNorm /
nn.Embedding submodules are used purely as named parameter containers; their forward is never called.
Layout: activations bf16 [B, T, d]; q/k/v of one attention are one fused GEMM ([k|v|q], arena order).
"""
import math

import contextlib

import torch
from torch import nn

from .. import kernels as K
from .. import ops
from .. import streams
from ..arena import ParamArena
from ..config import VacnicConfig
from ..ops import LinearSpec


def _spec(lin, trainable=True):
    w, b = lin.weight, lin.bias
    return LinearSpec(w.w16, b.data if b is not None else None, w.grad if trainable else None,
                      b.grad if (trainable and b is not None) else None)


class MLPClipCap(nn.Module):
    """MFULL:111-123 — Linear(cw -> 768P/2) -> Tanh -> Linear(768P/2 -> 768P); parameter names model.0 / model.2."""

    def __init__(self, sizes, bias=True):
        super().__init__()
        assert len(sizes) == 3
        self.model = nn.Sequential(nn.Linear(sizes[0], sizes[1], bias=bias), nn.Tanh(), nn.Linear(sizes[1], sizes[2], bias=bias))

    def bind_arena(self, arena):
        self.s0, self.s2 = _spec(self.model[0], arena.trainable), _spec(self.model[2], arena.trainable)

    def forward(self, x):
        return ops.mlp2(x, self.model[0].weight, self.s0, self.s2, act="tanh")


class MLP(nn.Module):
    """MFULL:76-108 (`--prompt_mlp_type mlp`): a token-mixing MLP over the ViT patch tokens.  The reference *reshapes*
    (not transposes) [B, n_tokens, width] to [B, width, n_tokens], runs Linear(+Tanh) over the last axis down to
    sizes[-1], and reshapes back to [B, sizes[-1], width]; parameter names model.0 / model.2 / model.4 ..."""

    def __init__(self, sizes, hidden_size=768, bias=True):
        super().__init__()
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(nn.Linear(sizes[i], sizes[i + 1], bias=bias))
            if i < len(sizes) - 2:
                layers.append(nn.Tanh())
        self.sizes = list(sizes)
        self.model = nn.Sequential(*layers)

    def bind_arena(self, arena):
        self.specs = [_spec(m, arena.trainable) for m in self.model if isinstance(m, nn.Linear)]

    def forward(self, x):
        if x.dim() != 3 or x.shape[1] != self.sizes[0]:
            raise ValueError(f"prompt MLP expects image_features [B, {self.sizes[0]}, width] (ln_post patch tokens, TRAIN:220-240), "
                             f"got {tuple(x.shape)}")
        B, feat, hidden = x.shape
        y = ops.mlp_chain(x.reshape(B * hidden, feat), self.model[0].weight, self.specs, act="tanh")
        return y.reshape(B, self.sizes[-1], hidden)


class BartLearnedPositionalEmbedding(nn.Embedding):
    """MFULL:401-418: table has 2 extra rows; position t reads row t+2 (done inside the embed kernels)."""

    def __init__(self, num_embeddings, embedding_dim):
        self.offset = 2
        super().__init__(num_embeddings + self.offset, embedding_dim)


class BartAttention(nn.Module):
    """MFULL:421-565.  forward(hidden_states, key_value_states=None, key_mask=None, causal=False) returns the
    out_proj output; the additive [B,1,T,S] mask of the reference is replaced by its generator: a per-key
    uint8 mask (0 = masked with finfo.min) and a causal flag."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, is_decoder=False, bias=True, cross_only=False):
        """cross_only: an attention that is only ever used with key_value_states (the decoder's encoder_attn): its k|v
        parameters are laid out by the owner (BartDecoder batches all layers' k|v into one arena group), q separately."""
        super().__init__()
        self.cross_only = cross_only
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        if self.head_dim * num_heads != embed_dim:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {embed_dim} and `num_heads`: {num_heads}).")
        if self.head_dim != 64:
            raise ValueError("the gfx950 attention kernels are built for head_dim 64 (bart-base/large, CLIP ViT)")
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"attention dropout must be in [0, 1), got {dropout}")
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = is_decoder
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)

    def arena_groups(self):
        if self.cross_only:
            return [[self.k_proj.weight, self.v_proj.weight], [self.k_proj.bias, self.v_proj.bias]]
        return [[self.k_proj.weight, self.v_proj.weight, self.q_proj.weight], [self.k_proj.bias, self.v_proj.bias, self.q_proj.bias]]

    def bind_arena(self, a):
        ws = [self.k_proj.weight, self.v_proj.weight, self.q_proj.weight]
        bs = [self.k_proj.bias, self.v_proj.bias, self.q_proj.bias]
        t = a.trainable
        if not self.cross_only:
            self.s_kvq = LinearSpec(a.fused(ws, "w16"), a.fused(bs, "f32"), a.fused(ws, "grad") if t else None, a.fused(bs, "grad") if t else None)
        self.s_kv = LinearSpec(a.fused(ws[:2], "w16"), a.fused(bs[:2], "f32"), a.fused(ws[:2], "grad") if t else None, a.fused(bs[:2], "grad") if t else None)
        self.s_q = _spec(self.q_proj, t)
        self.s_out = _spec(self.out_proj, t)

    def project_kv(self, key_value_states):
        """fused k|v projection of a cross-attention source (MFULL:478-484)."""
        return ops.linear(key_value_states, self.k_proj.weight, self.s_kv)

    def forward(self, hidden_states, key_value_states=None, key_mask=None, causal=False, kv=None, skip=False, bank=None, slot=0):
        """kv: optional precomputed project_kv(key_value_states).  skip=True: returns (output, hidden_states) — the residual
        branch of the block, whose gradient joins the input gradient inside the first projection's dgrad GEMM."""
        H = self.num_heads
        lin = ops.linear_skip if skip else ops.linear
        res = None
        if key_value_states is None and kv is None:
            kvq = lin(hidden_states, self.k_proj.weight, self.s_kvq)
            if skip:
                kvq, res = kvq
            ctx = ops.self_attention(kvq, key_mask, causal, H, self.dropout, self.training)                 # MFULL:546
        else:
            q = lin(hidden_states, self.q_proj.weight, self.s_q)
            if skip:
                q, res = q
            if kv is None:
                kv = self.project_kv(key_value_states)
            ctx = ops.cross_attention(q, kv, key_mask, H, bank, slot, self.dropout, self.training)
        out = ops.linear(ctx, self.out_proj.weight, self.s_out)
        return (out, res) if skip else out


class _Ffn(nn.Module):
    pass


class BartEncoderLayer(nn.Module):
    """MFULL:568-762 (only_image=True: MVIS:560-690)."""

    def __init__(self, config: VacnicConfig, visual_feature_dim=None, text_feature_dim=None, dim_common=None,
                 max_ner_type_len=80, max_ner_type_len_gt=20, only_image=False):
        super().__init__()
        d = self.embed_dim = config.d_model
        H = config.encoder_attention_heads
        self.self_attn = BartAttention(d, H, dropout=config.attention_dropout)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.dropout = config.dropout
        if config.activation_function != "gelu":
            raise NotImplementedError("activation_function must be gelu (bart-base/large)")
        self.activation_dropout = config.activation_dropout              # MFULL:580 (0.1 in the bart-base / bart-large hub configs: config.HUB_MODEL_DROPOUTS)
        self.fc1 = nn.Linear(d, config.encoder_ffn_dim)
        self.fc2 = nn.Linear(config.encoder_ffn_dim, d)
        self.final_layer_norm = nn.LayerNorm(d)
        self._linear_1up = nn.Linear(d, config.encoder_ffn_dim)
        self._linear_1down = nn.Linear(config.encoder_ffn_dim, d)
        self.img_layer_norm = nn.LayerNorm(d)
        self.only_image = only_image
        if not only_image:
            self.ner_map_up = nn.Linear(max_ner_type_len, 4 * max_ner_type_len_gt)
            self.ner_map_down = nn.Linear(4 * max_ner_type_len_gt, max_ner_type_len_gt)
            self.ner_map_layer_norm = nn.LayerNorm(d)
            self.max_ner_type_len_gt = max_ner_type_len_gt
            self.self_attn_img_name = BartAttention(d, H, dropout=config.attention_dropout)
            self.img_name_attn_layer_norm = nn.LayerNorm(d)
            self._face_up = nn.Linear(d, 3072)                     # hard-coded 3072, MFULL:607
            self._face_down = nn.Linear(3072, d)
            self.face_layer_norm = nn.LayerNorm(d)
        self.cross_attn_img_ner = BartAttention(d, H, dropout=config.attention_dropout)
        self.img_ner_attn_layer_norm = nn.LayerNorm(d)

    def bind_arena(self, a):
        t = a.trainable
        self.s_fc1, self.s_fc2 = _spec(self.fc1, t), _spec(self.fc2, t)
        self.s_up, self.s_down = _spec(self._linear_1up, t), _spec(self._linear_1down, t)
        if not self.only_image:
            self.s_nup, self.s_ndown = _spec(self.ner_map_up, t), _spec(self.ner_map_down, t)
            self.s_fup, self.s_fdown = _spec(self._face_up, t), _spec(self._face_down, t)

    def _ln(self, x, res, ln, drop=True):
        return ops.add_ln(x, res, ln.weight, ln.bias, self.dropout if drop else 0.0, self.training)

    def forward(self, hidden_states, key_mask, hidden_states_img=None, hidden_states_face=None, hidden_states_ner=None,
                face_name_key_mask=None, add_ner_ffn=True, fused=True, on_branch=False):
        """on_branch (explicit scheduling only): the image / face / name inputs were produced on the branch stream by an earlier
        layer; the caller (BartEncoder) hops the last layer's outputs back to the compute stream."""
        h = hidden_states
        if fused:
            # The image / face / name branches (MFULL:647-691) are a dozen small kernels over 20-80 tokens per sample and are
            # independent of the text self-attention block of this layer; with side streams on (training only) they run on
            # the branch stream beside it and join before the cross-attention that consumes their output.
            br = streams.branch_stream() if (torch.is_grad_enabled() and hidden_states.is_cuda) else None
            expl = br is not None and streams.explicit()
            if expl:
                # explicit scheduling: the branch ops are launched on the branch stream through kernels.launch_on, and the tensors
                # that cross between the two streams go through ops.stream_hop (fence forward, mirrored fence in backward)
                main_raw, br_raw = K._stream(), streams.raw("branch")
                if not on_branch:
                    ins = [t for t in (hidden_states_img, hidden_states_face, hidden_states_ner) if t is not None]
                    outs = list(ops.stream_hop(main_raw, br_raw, *ins))
                    hidden_states_img = outs.pop(0)
                    if hidden_states_face is not None:
                        hidden_states_face = outs.pop(0)
                    if hidden_states_ner is not None:
                        hidden_states_ner = outs.pop(0)
                branch_ctx = K.launch_on(br_raw)
            elif br is not None:
                cur = torch.cuda.current_stream()
                br.wait_stream(cur)
                branch_ctx = torch.cuda.stream(br)
            else:
                branch_ctx = contextlib.nullcontext()
            with branch_ctx:
                # img FFN (MFULL:647-653)
                a, r = ops.mlp2_skip(hidden_states_img, self.fc1.weight, self.s_up, self.s_down, p_act=self.activation_dropout, training=self.training)   # :649
                hidden_states_img = self._ln(a, r, self.img_layer_norm)
                if not self.only_image:
                    if not add_ner_ffn:
                        # MFULL:665-666 concatenates [img ; names ; text] as keys while the mask of MFULL:1293-1296 covers
                        # P + max_ner_type_len_gt of them: the reference's own BartAttention rejects the call (MFULL:520-524;
                        # verified by running the reference).  Same error, same wording.
                        Bq, Sq = hidden_states.shape[0], hidden_states.shape[1]
                        n_kv = hidden_states_img.shape[1] + hidden_states_ner.shape[1] + Sq
                        n_mask = hidden_states_img.shape[1] + self.max_ner_type_len_gt
                        raise ValueError(f"Attention mask should be of size {(Bq, 1, Sq, n_kv)}, but is "
                                         f"torch.Size([{Bq}, 1, {Sq}, {n_mask}])")
                    # face FFN (:658-664)
                    a, r = ops.mlp2_skip(hidden_states_face, self.fc1.weight, self.s_fup, self.s_fdown, p_act=self.activation_dropout, training=self.training)   # :660
                    hidden_states_face = self._ln(a, r, self.face_layer_norm)
                    face_kv, face_out = ops.fork(hidden_states_face)
                    # names attend to [faces ; names] (:666-679) — no dropout on this branch in the reference
                    ner_q, ner_r = ops.fork(hidden_states_ner)
                    ner_q, ner_k = ops.fork(ner_q)
                    kv_src = ops.cat_tokens(face_kv, ner_k)
                    att = self.self_attn_img_name(ner_q, key_value_states=kv_src, key_mask=face_name_key_mask)
                    hidden_states_ner = self._ln(att, ner_r, self.img_name_attn_layer_norm, drop=False)
                    ner_p, ner_out = ops.fork(hidden_states_ner)
                    # name-prefix FFN on the FLAT view [B, d, N] (reshape, not transpose; :682-688)
                    B, N, d = ner_p.shape
                    pre = ops.mlp2(ner_p.reshape(B, d, N), self.fc1.weight, self.s_nup, self.s_ndown, p_act=self.activation_dropout, training=self.training)   # :684
                    pre = pre.reshape(B, self.max_ner_type_len_gt, d)
                    pre = ops.add_ln(pre, None, self.ner_map_layer_norm.weight, self.ner_map_layer_norm.bias, self.dropout, self.training)
                    img_kv, img_out = ops.fork(hidden_states_img)
                    kv = ops.cat_tokens(img_kv, pre)                                               # :691
                    hidden_states_img, hidden_states_face, hidden_states_ner = img_out, face_out, ner_out
                else:
                    kv, hidden_states_img = ops.fork(hidden_states_img)
                kv = self.cross_attn_img_ner.project_kv(kv)        # k|v projection of the [img ; prefix] tokens: also off the text chain
            a, r = self.self_attn(h, key_mask=key_mask, skip=True)
            h = self._ln(a, r, self.self_attn_layer_norm)                                          # :697-707
            if expl:
                (kv,) = ops.stream_hop(br_raw, main_raw, kv)
            elif br is not None:
                cur.wait_stream(br)
            a, r = self.cross_attn_img_ner(h, kv=kv, key_mask=None, skip=True)
            h = self._ln(a, r, self.img_ner_attn_layer_norm)                                       # :711-723
        else:
            a, r = self.self_attn(h, key_mask=key_mask, skip=True)
            h = self._ln(a, r, self.self_attn_layer_norm)                                          # :726-736
        a, r = ops.mlp2_skip(h, self.fc1.weight, self.s_fc1, self.s_fc2, p_act=self.activation_dropout, training=self.training)   # :740
        h = self._ln(a, r, self.final_layer_norm)                                                  # :738-744
        return h, hidden_states_face, hidden_states_ner, hidden_states_img


class BartDecoderLayer(nn.Module):
    """MFULL:765-890 (training / no-cache path)."""

    def __init__(self, config: VacnicConfig):
        super().__init__()
        d = self.embed_dim = config.d_model
        self.self_attn = BartAttention(d, config.decoder_attention_heads, dropout=config.attention_dropout, is_decoder=True)
        self.dropout = config.dropout
        self.activation_dropout = config.activation_dropout              # MFULL:778
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.encoder_attn = BartAttention(d, config.decoder_attention_heads, dropout=config.attention_dropout, is_decoder=True, cross_only=True)
        self.encoder_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, config.decoder_ffn_dim)
        self.fc2 = nn.Linear(config.decoder_ffn_dim, d)
        self.final_layer_norm = nn.LayerNorm(d)

    def bind_arena(self, a):
        self.s_fc1, self.s_fc2 = _spec(self.fc1, a.trainable), _spec(self.fc2, a.trainable)

    def _ln(self, x, res, ln):
        return ops.add_ln(x, res, ln.weight, ln.bias, self.dropout, self.training)

    def forward(self, hidden_states, encoder_hidden_states, encoder_key_mask, kv=None, bank=None, slot=0):
        a, r = self.self_attn(hidden_states, causal=hidden_states.shape[1] > 1, skip=True)
        h = self._ln(a, r, self.self_attn_layer_norm)
        a, r = self.encoder_attn(h, key_value_states=encoder_hidden_states, key_mask=encoder_key_mask, kv=kv, skip=True, bank=bank, slot=slot)
        h = self._ln(a, r, self.encoder_attn_layer_norm)
        a, r = ops.mlp2_skip(h, self.fc1.weight, self.s_fc1, self.s_fc2, p_act=self.activation_dropout, training=self.training)   # MFULL:874
        return self._ln(a, r, self.final_layer_norm)


class BartEncoder(nn.Module):
    """MFULL:1089-1381 (MVIS:1017-1251)."""

    def __init__(self, config: VacnicConfig, embed_tokens=None, fusion_layer=None, dim_common=256, face_visual_feature_dim=512,
                 img_size=2048, prompt_mlp_type="clipcap", map_size=None, prompt_size=10, max_ner_type_len=80,
                 max_ner_type_len_gt=20, only_image=False):
        super().__init__()
        self.config = config
        self.dropout = config.dropout
        d = self.embed_dim = config.d_model
        self.padding_idx = config.pad_token_id
        self.embed_scale = math.sqrt(d) if config.scale_embedding else 1.0
        self.embed_tokens = embed_tokens if embed_tokens is not None else nn.Embedding(config.vocab_size, d, self.padding_idx)
        self.embed_positions = BartLearnedPositionalEmbedding(config.max_position_embeddings, d)
        self.layers = nn.ModuleList([BartEncoderLayer(config, img_size, d, dim_common, max_ner_type_len, max_ner_type_len_gt, only_image)
                                     for _ in range(config.encoder_layers)])
        self.layernorm_embedding = nn.LayerNorm(d)
        self.fusion_layer = list(fusion_layer or [])
        if prompt_mlp_type == "clipcap":
            self.prompt_mlp = MLPClipCap((config.clip_width, (768 * prompt_size) // 2, 768 * prompt_size))   # MFULL:1136 (768 -> clip_width)
        else:
            map_size = list(map_size if map_size is not None else (config.map_size or [192, 256, 64, 16]))   # MFULL:1099 default
            self.prompt_mlp = MLP(map_size, hidden_size=768)                                                # MFULL:1138
        self.prompt_size, self.prompt_mlp_type = prompt_size, prompt_mlp_type
        if d == 1024:
            self.visual_map = nn.Linear(768, 1024)
        self.only_image = only_image
        if not only_image:
            self.embed_tokens_ner = nn.Embedding(50267, d, self.padding_idx)
            self.embed_positions_ner = BartLearnedPositionalEmbedding(config.max_position_embeddings, d)
            self.layernorm_embedding_ner = nn.LayerNorm(d)
        self.max_ner_type_len, self.max_ner_type_len_gt = max_ner_type_len, max_ner_type_len_gt
        self._linear_1 = nn.Linear(face_visual_feature_dim, dim_common)

    def bind_arena(self, a):
        self.s_l1 = _spec(self._linear_1, a.trainable)
        if self.embed_dim == 1024:
            self.s_vmap = _spec(self.visual_map, a.trainable)

    def get_input_embeddings(self):
        return self.embed_tokens

    def forward(self, input_ids=None, attention_mask=None, image_features=None, name_ids=None, name_mask=None,
                face_features=None, face_mask=None, add_ner_ffn=True, **unused):
        if input_ids is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        B, S = input_ids.shape
        key_mask = attention_mask.to(torch.uint8) if attention_mask.dtype != torch.uint8 else attention_mask
        ln = self.layernorm_embedding
        h = ops.embed_ln(input_ids, self.embed_tokens.weight, self.embed_positions.weight, ln.weight, ln.bias, self.embed_scale,
                         self.dropout, self.training, self.padding_idx)                                   # :1243-1249
        face = ner = fn_mask = None
        if not self.only_image:
            if name_ids.shape[1] != self.max_ner_type_len:
                raise ValueError(f"name_ids length {name_ids.shape[1]} must equal max_ner_type_len={self.max_ner_type_len} (MFULL:595,683)")
            ln = self.layernorm_embedding_ner
            ner = ops.embed_ln(name_ids, self.embed_tokens_ner.weight, self.embed_positions_ner.weight, ln.weight, ln.bias,
                               self.embed_scale, self.dropout, self.training, self.padding_idx)           # :1254-1260
            fn_mask = K.cat_masks(face_mask.to(torch.uint8).contiguous(), name_mask.to(torch.uint8).contiguous())   # :1262 (mask bytes only)
            face = ops.linear(ops.to_bf16(face_features), self._linear_1.weight, self.s_l1)              # :1269
        img = self.prompt_mlp(ops.to_bf16(image_features))                                               # :1274
        if self.prompt_mlp_type == "clipcap":
            img = img.reshape(B, self.prompt_size, 768)                                                   # :1275-1276
        if self.embed_dim == 1024:
            img = ops.linear(img, self.visual_map.weight, self.s_vmap)                                   # :1277-1278
        # img_ner_mask_cross is all ones (:1280-1296) -> no key mask on the visual/name cross-attention
        # explicit scheduling: from the first fusion layer on the image / face / name streams live on the branch stream
        expl = streams.explicit() and streams.branch_stream() is not None and torch.is_grad_enabled() and h.is_cuda
        on_branch = False
        for idx, layer in enumerate(self.layers):
            fused = idx in self.fusion_layer
            h, face, ner, img = layer(h, key_mask, hidden_states_img=img, hidden_states_face=face, hidden_states_ner=ner,
                                      face_name_key_mask=fn_mask, add_ner_ffn=add_ner_ffn, fused=fused, on_branch=on_branch)
            on_branch = on_branch or (fused and expl)
        if on_branch:                               # back to the compute stream (SECLA reads the face stream; TRAIN:326-330)
            outs = list(ops.stream_hop(streams.raw("branch"), K._stream(), *[t for t in (img, face, ner) if t is not None]))
            img = outs.pop(0)
            if face is not None:
                face = outs.pop(0)
            if ner is not None:
                ner = outs.pop(0)
        return {"last_hidden_state": h, "hidden_states_img": img, "hidden_states_ner": ner, "hidden_states_face": face}


class BartDecoder(nn.Module):
    """MFULL:1384-1675 (training / teacher-forced path; decoder_attention_mask=None as at TRAIN:281)."""

    def __init__(self, config: VacnicConfig, embed_tokens=None):
        super().__init__()
        self.config = config
        self.dropout = config.dropout
        d = self.embed_dim = config.d_model
        self.padding_idx = config.pad_token_id
        self.embed_scale = math.sqrt(d) if config.scale_embedding else 1.0
        self.embed_tokens = embed_tokens if embed_tokens is not None else nn.Embedding(config.vocab_size, d, self.padding_idx)
        self.embed_positions = BartLearnedPositionalEmbedding(config.max_position_embeddings, d)
        self.layers = nn.ModuleList([BartDecoderLayer(config) for _ in range(config.decoder_layers)])
        self.layernorm_embedding = nn.LayerNorm(d)

    def arena_groups(self):
        """all layers' cross-attention k|v parameters adjacent ([k0, v0, k1, v1, ...]): ONE GEMM X[B*S, d] . W[L*2d, d]^T projects the
        encoder output for every layer (the reference recomputes k_proj / v_proj of the same states per layer, MFULL:478-484),
        and its backward is one dgrad with K = L*2d and one weight gradient."""
        ats = [l.encoder_attn for l in self.layers]
        return [[p for a in ats for p in (a.k_proj.weight, a.v_proj.weight)], [p for a in ats for p in (a.k_proj.bias, a.v_proj.bias)]]

    def bind_arena(self, a):
        ws, bs = self.arena_groups()
        t = a.trainable
        self.s_kv_all = LinearSpec(a.fused(ws, "w16"), a.fused(bs, "f32"), a.fused(ws, "grad") if t else None, a.fused(bs, "grad") if t else None)

    def forward(self, input_ids, encoder_hidden_states, encoder_attention_mask, output_hidden_states=True):
        ln = self.layernorm_embedding
        h = ops.embed_ln(input_ids, self.embed_tokens.weight, self.embed_positions.weight, ln.weight, ln.bias, self.embed_scale,
                         self.dropout, self.training, self.padding_idx)
        enc_mask = encoder_attention_mask.to(torch.uint8) if encoder_attention_mask.dtype != torch.uint8 else encoder_attention_mask
        states = [h] if output_hidden_states else None
        n = len(self.layers)
        kv_all = ops.linear(encoder_hidden_states, self.layers[0].encoder_attn.k_proj.weight, self.s_kv_all)     # [B, S, n*2d]
        kvs, bank = ops.split_kv(kv_all, n)
        for i, layer in enumerate(self.layers):
            h = layer(h, None, enc_mask, kv=kvs[i], bank=bank, slot=i)
            if output_hidden_states:
                h, keep = ops.fork(h)          # every state has two consumers (next layer / lm_head, and the caller)
                states.append(keep)
        return h, states


class BartModel(nn.Module):
    """MFULL:1702-1855."""

    def __init__(self, config: VacnicConfig, enc_fusion_layer=None, dim_common=256, img_size=2048, prompt_mlp_type="clipcap",
                 map_size=None, prompt_size=10, max_ner_type_len=80, max_ner_type_len_gt=20, only_image=False):
        super().__init__()
        self.config = config
        self.shared = nn.Embedding(config.vocab_size, config.d_model, config.pad_token_id)
        self.encoder = BartEncoder(config, self.shared, fusion_layer=enc_fusion_layer, dim_common=dim_common, img_size=img_size,
                                   prompt_mlp_type=prompt_mlp_type, map_size=map_size, prompt_size=prompt_size,
                                   max_ner_type_len=max_ner_type_len, max_ner_type_len_gt=max_ner_type_len_gt, only_image=only_image)
        self.decoder = BartDecoder(config, self.shared)
        self.embed_dim = config.d_model

    def get_input_embeddings(self):
        return self.shared

    def get_encoder(self):
        return self.encoder

    def get_decoder(self):
        return self.decoder

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, image_features=None, face_features=None,
                face_mask=None, name_ids=None, name_mask=None, add_ner_ffn=True, encoder_outputs=None, **unused):
        if decoder_input_ids is None:
            if input_ids is None:
                raise ValueError("If no `decoder_input_ids` or `decoder_inputs_embeds` are passed, `input_ids` cannot be `None`.")
            _, decoder_input_ids = K.prep_ids(input_ids, self.config.pad_token_id, self.config.decoder_start_token_id, want_mask=False)
        if encoder_outputs is None:
            encoder_outputs = self.encoder(input_ids=input_ids, attention_mask=attention_mask, image_features=image_features,
                                           name_ids=name_ids, name_mask=name_mask, face_features=face_features,
                                           face_mask=face_mask, add_ner_ffn=add_ner_ffn)
        h, states = self.decoder(decoder_input_ids, encoder_outputs["last_hidden_state"], attention_mask)
        return {"last_hidden_state": h, "decoder_hidden_states": tuple(states),
                "encoder_last_hidden_state": encoder_outputs["last_hidden_state"],
                "hidden_states_face": encoder_outputs["hidden_states_face"], "hidden_states_ner": encoder_outputs["hidden_states_ner"],
                "hidden_states_img": encoder_outputs["hidden_states_img"]}


def init_attn_weight_encoder(encoder):
    """MFULL:1858-1870 (`--init_attn_weight True`): in every encoder layer the q/k/v/out projection WEIGHTS (not biases) of the
    name self-attention and of the image/name cross-attention become the text self-attention's Parameter objects — a true tie
    for the whole run, so each weight's gradient is the sum over its three uses.  In the arena the tied attentions read the
    same bf16 shadow and accumulate into the same gradient view (their weight-gradient GEMMs are serialized on one stream).
    Like the reference, a layer without `self_attn_img_name` (only_image) raises AttributeError."""
    for layer in encoder.layers:
        for tied in (layer.self_attn_img_name, layer.cross_attn_img_ner):
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                getattr(tied, proj).weight = getattr(layer.self_attn, proj).weight


class BartForMultiModalGeneration(nn.Module):
    """MFULL:1877-2074 (MVIS:1731-1914 when only_image=True).  forward() returns the dict the trainer indexes
    (`logits`, `decoder_hidden_states`, `hidden_states_face`, ...; TRAIN:281-294,326).

    Extra (MI355X-native) entry: `forward(..., labels=tgt_ids)` fuses lm_head + CrossEntropyLoss(ignore_index=pad)
    and returns `loss` without exposing logits (the reference computes it at TRAIN:287 from materialised logits).
    Call `.finalize(device)` once after loading weights: it moves all parameters into the flat arena."""

    def __init__(self, config: VacnicConfig, enc_fusion_layer=None, dim_common=256, img_size=2048, prompt_mlp_type="clipcap",
                 map_size=None, prompt_size=10, clip_model=None, freeze_clip=False, max_ner_type_len=80, max_ner_type_len_gt=20,
                 only_image=False, init_attn_weight=False):
        super().__init__()
        config.validate()
        self.config = config
        self.model = BartModel(config, enc_fusion_layer, dim_common, img_size, prompt_mlp_type, map_size, prompt_size,
                               max_ner_type_len, max_ner_type_len_gt, only_image)
        V = self.model.shared.num_embeddings
        self.register_buffer("final_logits_bias", torch.zeros((1, V)))
        self.lm_head = nn.Linear(config.d_model, V, bias=False)
        self.lm_head.weight = self.model.shared.weight                 # tie (HF tie_weights; MFULL:1885)
        self.clip_model = clip_model
        if freeze_clip and clip_model is not None:
            for p in clip_model.parameters():
                p.requires_grad = False
        if init_attn_weight or config.init_attn_weight:
            init_attn_weight_encoder(self.model.encoder)
        self._init_weights()
        self.arena = None

    def _init_weights(self):
        std = self.config.init_std
        for m in self.model.modules():                                  # BartPretrainedModel._init_weights, MFULL:899-908
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=std)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(mean=0.0, std=std)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()

    def get_encoder(self):
        return self.model.get_encoder()

    def get_decoder(self):
        return self.model.get_decoder()

    def get_output_embeddings(self):
        return self.lm_head

    def trainable_parameters(self):
        """list(model.model.parameters()) + list(model.lm_head.parameters()) of TRAIN:91, deduplicated: the tied
        lm_head/shared matrix gets ONE update per step (the reference's duplicate entry is an accident)."""
        return list(self.model.parameters())

    def finalize(self, device="cuda", trainable=True):
        V = self.model.shared.num_embeddings
        self.V, self.V_pad = V, (V + 31) // 32 * 32
        self.final_logits_bias = self.final_logits_bias.to(device)
        self.arena = ParamArena(self.model, device, trainable=trainable, pad_rows={id(self.model.shared.weight): self.V_pad})
        self.emb16_pad = self.arena.view16(self.model.shared.weight, rows=self.V_pad)
        self.s_lm = LinearSpec(self.model.shared.weight.w16, self.final_logits_bias.view(-1),
                               self.model.shared.weight.grad if trainable else None, None)
        return self

    def resize_token_embeddings(self, new_num_tokens):
        if self.arena is not None:
            raise RuntimeError("resize_token_embeddings must be called before finalize()")
        old = self.model.shared
        if new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim, old.padding_idx)
        new.weight.data.normal_(mean=0.0, std=self.config.init_std)
        n = min(new_num_tokens, old.num_embeddings)
        new.weight.data[:n] = old.weight.data[:n]
        self.model.shared = new
        self.model.encoder.embed_tokens = new
        self.model.decoder.embed_tokens = new
        self.lm_head = nn.Linear(old.embedding_dim, new_num_tokens, bias=False)
        self.lm_head.weight = new.weight
        fb = torch.zeros((1, new_num_tokens))
        fb[:, :min(new_num_tokens, self.final_logits_bias.shape[1])] = self.final_logits_bias[:, :new_num_tokens]
        self.final_logits_bias = fb
        self.config.vocab_size = new_num_tokens
        return new

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, image_features=None, face_features=None,
                face_mask=None, name_ids=None, name_mask=None, add_ner_ffn=True, labels=None, output_logits=None,
                encoder_outputs=None, **unused):
        if self.arena is None:
            raise RuntimeError("call model.finalize(device) before forward (parameters must live in the HBM arena)")
        if labels is not None and decoder_input_ids is None:
            _, decoder_input_ids = K.prep_ids(labels, self.config.pad_token_id, self.config.decoder_start_token_id, want_mask=False)
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, decoder_input_ids=decoder_input_ids,
                         image_features=image_features, face_features=face_features, face_mask=face_mask, name_ids=name_ids,
                         name_mask=name_mask, add_ner_ffn=add_ner_ffn, encoder_outputs=encoder_outputs)
        h = out.pop("last_hidden_state")
        want_logits = output_logits if output_logits is not None else labels is None
        if labels is not None:
            hl, h = ops.fork(h) if want_logits else (h, None)
            loss, acc = ops.lm_head_ce(hl, self.model.shared.weight, self.emb16_pad, self.model.shared.weight.grad, labels,
                                       self.V, self.config.pad_token_id)
            out["loss"], out["loss_acc"] = loss, acc
        if want_logits:
            lg = ops.linear(h, self.model.shared.weight, self.s_lm)                           # lm_head(h) + final_logits_bias, MFULL:1997
            out["logits"] = lg
        return out

    # ---- decoding (TRAIN:480-559, DDPINF:758-842 call model.generate) ----
    def generate(self, input_ids=None, attention_mask=None, **kw):
        """KV-cached greedy / beam search with transformers-4.18 semantics: see vacnic_amd/generate.py."""
        from ..generate import generate as _generate
        return _generate(self, input_ids=input_ids, attention_mask=attention_mask, **kw)

    def prepare_inputs_for_generation(self, decoder_input_ids, past=None, **kw):
        """MFULL:2023-2061 contract: with a cache only the last token is fed (the cached decoder does exactly that)."""
        if past is not None:
            decoder_input_ids = decoder_input_ids[:, -1:]
        return dict(kw, decoder_input_ids=decoder_input_ids, past_key_values=past)

    @torch.no_grad()
    def greedy_generate(self, input_ids, attention_mask, max_length, **kw):
        enc = self.model.encoder(input_ids=input_ids, attention_mask=attention_mask, **kw)
        B = input_ids.shape[0]
        ids = torch.full((B, 1), self.config.decoder_start_token_id, dtype=torch.long, device=input_ids.device)
        for _ in range(max_length - 1):
            h, _ = self.model.decoder(ids, enc["last_hidden_state"], attention_mask, output_hidden_states=False)
            last = h[:, -1].contiguous()
            lg = ops.linear(last, self.model.shared.weight, self.s_lm)
            nxt = K.argmax_rows(lg, self.V)
            ids = torch.cat([ids, nxt[:, None]], dim=1)
        return ids
