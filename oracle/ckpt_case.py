"""TEST INFRASTRUCTURE (never imported by the product): the seeded checkpoint of the "read back by the reference" fixture
(SURVEY §8f-3).  `make_checkpoint()` builds the product's model on host arenas, writes a checkpoint with
vacnic_amd/checkpoint.py exactly as the trainer does, and returns the loaded dict plus the inputs of the fixture's forward;
oracle/make_golden.py loads `ck["model"]` into the REAL reference class (strict) and records its logits,
tests/test_oracle.py replays the same checkpoint through the oracle against that record."""
import io

import torch

from vacnic_amd import checkpoint, synthetic
from vacnic_amd.config import VacnicConfig
from vacnic_amd.models.mmbart import BartForMultiModalGeneration
from vacnic_amd.training import FusedAdamW, load_named


def case_cfg():
    return VacnicConfig(d_model=768, encoder_layers=1, decoder_layers=1, encoder_attention_heads=12, decoder_attention_heads=12,
                        encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=768, dropout=0.0)


def make_checkpoint():
    cfg = case_cfg()
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=17)
    m = BartForMultiModalGeneration(cfg, enc_fusion_layer=cfg.enc_fusion_layer, dim_common=cfg.dim_common, img_size=768,
                                    prompt_mlp_type=cfg.prompt_mlp_type, prompt_size=cfg.prompt_size, max_ner_type_len=cfg.max_ner_type_len,
                                    max_ner_type_len_gt=cfg.max_ner_type_len_gt, only_image=cfg.only_image)
    load_named(m, sd)
    m.finalize("cpu")
    opt = FusedAdamW(m.arena, lr=3e-5, num_warmup_steps=2, num_training_steps=40)
    buf = io.BytesIO()
    checkpoint.save_checkpoint(buf, m, opt, step=3, config=dict(cfg.__dict__))
    buf.seek(0)
    ck = torch.load(buf, map_location="cpu", weights_only=False)
    batch = synthetic.make_batch(cfg, 2, S=24, T=8, F=2, seed=19, image_size=32)
    img = synthetic._normal("img_cls", (2, 768), 1.0, 23)
    return cfg, ck, batch, img
