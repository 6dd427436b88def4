"""__graft_entry__.smoke(): one tiny VACNIC train step on cuda:0 through the HIP kernels, checked against the
CPU oracle's loss on the same seeded weights and batch."""
import os
import sys

import torch


def run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import vacnic_oracle as O            # checker only
    from . import synthetic
    from .config import ClipVisionConfig, VacnicConfig
    from .training import FusedAdamW, TrainArgs, build_models, forward_losses, to_device, train_step
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    torch.cuda.set_device(0)
    cfg = VacnicConfig(d_model=768, encoder_layers=1, decoder_layers=1, encoder_attention_heads=12, decoder_attention_heads=12,
                       encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=128,
                       dropout=0.0).validate()
    vcfg = ClipVisionConfig(width=128, layers=1, patch_size=16, image_size=32, output_dim=64)
    model, guide, clip_model = build_models(cfg, vcfg, init="synthetic", seed=0)
    batch = synthetic.make_batch(cfg, 2, S=32, T=8, F=2, seed=5, image_size=32)
    args = TrainArgs(num_training_steps=10)
    opt = FusedAdamW(model.arena, lr=args.lr_bart, num_warmup_steps=1, num_training_steps=10)
    model.eval()
    with torch.no_grad():
        _, out4, _ = forward_losses(model, guide, to_device(batch, "cuda"), args)
    sd = synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=1)
    sd_g = synthetic.make_state_dict(synthetic.guide_bart_param_shapes(cfg), seed=2)
    sd_c = synthetic.make_state_dict(synthetic.clip_visual_param_shapes(vcfg), seed=4, std=0.05)
    with torch.no_grad():
        ref = O.train_losses(sd, sd_g, sd_c, cfg, vcfg, batch, margin=args.margin, alpha=args.alpha,
                             mapping_loss_weight=args.mapping_loss_weight)
    got, want = out4[0].item(), ref["loss"].item()
    assert abs(got - want) <= 1e-2 * abs(want), f"HIP loss {got} vs oracle {want}"
    out = train_step(model, guide, opt, to_device(batch, "cuda"), args)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    print(f"smoke ok: loss {got:.4f} (oracle {want:.4f}); after one step {out.tolist()}")
