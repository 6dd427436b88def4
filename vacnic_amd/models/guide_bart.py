"""Frozen guide BART (CoLaM teacher): `BartForConditionalGeneration.from_pretrained(plm, output_hidden_states=True)`
with requires_grad=False (TRAIN:745-751), used once per step at TRAIN:293-294.  Same parameter names as the HF
class (model.shared / model.encoder.layers.N.self_attn... / model.decoder...), vanilla post-LN layers built from the
same gfx950 kernels as the trainable model.  Only decoder_hidden_states[-1] is consumed by the trainer, so the
LM head is skipped."""
import math

import torch
from torch import nn

from .. import kernels as K
from .. import ops
from ..arena import ParamArena
from ..config import VacnicConfig
from .mmbart import BartAttention, BartDecoder, BartLearnedPositionalEmbedding, _spec


class VanillaEncoderLayer(nn.Module):
    def __init__(self, config: VacnicConfig):
        super().__init__()
        d = config.d_model
        self.self_attn = BartAttention(d, config.encoder_attention_heads)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, config.encoder_ffn_dim)
        self.fc2 = nn.Linear(config.encoder_ffn_dim, d)
        self.final_layer_norm = nn.LayerNorm(d)

    def bind_arena(self, a):
        self.s_fc1, self.s_fc2 = _spec(self.fc1, a.trainable), _spec(self.fc2, a.trainable)

    def forward(self, h, key_mask):
        ln = self.self_attn_layer_norm
        h = ops.add_ln(self.self_attn(h, key_mask=key_mask), h, ln.weight, ln.bias)
        ln = self.final_layer_norm
        return ops.add_ln(ops.mlp2(h, self.fc1.weight, self.s_fc1, self.s_fc2), h, ln.weight, ln.bias)


class VanillaEncoder(nn.Module):
    def __init__(self, config, embed_tokens):
        super().__init__()
        d = config.d_model
        self.embed_tokens = embed_tokens
        self.embed_positions = BartLearnedPositionalEmbedding(config.max_position_embeddings, d)
        self.layers = nn.ModuleList([VanillaEncoderLayer(config) for _ in range(config.encoder_layers)])
        self.layernorm_embedding = nn.LayerNorm(d)
        self.embed_scale = math.sqrt(d) if config.scale_embedding else 1.0

    def forward(self, input_ids, key_mask):
        ln = self.layernorm_embedding
        h = ops.embed_ln(input_ids, self.embed_tokens.weight, self.embed_positions.weight, ln.weight, ln.bias, self.embed_scale)
        for layer in self.layers:
            h = layer(h, key_mask)
        return h


class _GuideModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.shared = nn.Embedding(config.vocab_size, config.d_model, config.pad_token_id)
        self.encoder = VanillaEncoder(config, self.shared)
        self.decoder = BartDecoder(config, self.shared)


class BartForConditionalGeneration(nn.Module):
    def __init__(self, config: VacnicConfig):
        super().__init__()
        self.config = config
        self.model = _GuideModel(config)
        std = config.init_std
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, std); m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0.0, std)
        for p in self.parameters():
            p.requires_grad = False
        self.arena = None
        self.eval()

    def finalize(self, device="cuda"):
        self.arena = ParamArena(self.model, device, trainable=False)
        return self

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, **unused):
        key_mask = attention_mask.to(torch.uint8) if attention_mask.dtype != torch.uint8 else attention_mask
        enc = self.model.encoder(input_ids, key_mask)
        h, states = self.model.decoder(decoder_input_ids, enc, key_mask, output_hidden_states=False)
        return {"decoder_hidden_states": (h,), "encoder_last_hidden_state": enc}
