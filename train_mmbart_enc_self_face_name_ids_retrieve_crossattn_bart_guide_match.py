"""Entry point with the reference trainer's file name and flags (TRAIN:5-82) for the MI355X-native VACNIC step.

    torchrun --nproc_per_node=N train_mmbart_enc_self_face_name_ids_retrieve_crossattn_bart_guide_match.py \
        --plm_type facebook/bart-large --clip_type ViT-L/14 --enc_fusion_layer 0 1 ... 11 --dim_common 1024 \
        --train_batch_size 32 --prompt_size 20 --use_secla True --margin 1.0 --alpha 0.5 --no_clip_norm True ...

Scope: the data-parallel training hot path.  Datasets, tokenizers, pretrained checkpoints, wandb and
caption scoring are out of scope and there is no network here, so `--data_type synthetic` (default) feeds
GoodNews-shaped synthetic batches and weights are random-initialised; a user with the real datasets
plugs a DataLoader that yields the same batch dict (DSG:22-127) into `run()`.
One process per GPU: rank/world come from torchrun's env (TRAIN:616-620 uses LOCAL_RANK the same way).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

_b = lambda x: (str(x).lower() == "true")    # noqa: E731  (TRAIN's boolean flag idiom)
parser = argparse.ArgumentParser()
parser.add_argument("--seed", type=str, default="684331")
parser.add_argument("--gpu_ids", type=str, default="0")
parser.add_argument("--num_workers", type=int, default=4)
parser.add_argument("--article_max_length", type=int, default=512)
parser.add_argument("--caption_max_length", type=int, default=100)
parser.add_argument("--plm_type", type=str, default="facebook/bart-large")
parser.add_argument("--clip_type", type=str, default="ViT-L/14")
parser.add_argument("--ent_start_token", type=str, default="no")
parser.add_argument("--ent_end_token", type=str, default="no")
parser.add_argument("--enc_fusion_layer", nargs="+", type=int)
parser.add_argument("--dim_common", type=int, default=1024)
parser.add_argument("--warmup_rate", type=float, default=0.05)
parser.add_argument("--train_batch_size", type=int, default=32)
parser.add_argument("--val_batch_size", type=int, default=1)
parser.add_argument("--test_batch_size", type=int, default=1)
parser.add_argument("--beam_size", type=int, default=1)
parser.add_argument("--max_length", type=int, default=50)
parser.add_argument("--num_epoch", type=int, default=1)
parser.add_argument("--lr_bart", type=float, default=3e-5)
parser.add_argument("--lr_clip", type=float, default=5e-6)
parser.add_argument("--weight_decay", type=float, default=0.01)
parser.add_argument("--clip_norm", type=float, default=0.1)
parser.add_argument("--data_type", type=str, default="synthetic")
parser.add_argument("--data_dir", type=str, default=".")
parser.add_argument("--out_dir", type=str, default=".")
parser.add_argument("--mapping_loss_type", type=str, default="contrastive")
parser.add_argument("--trained_clip", type=str, default="no")
parser.add_argument("--clip_dir", type=str, default=".")
parser.add_argument("--no_clip_loss", default=True, type=_b)
parser.add_argument("--prompt_size", type=int, default=20)
parser.add_argument("--use_vis_cls", default=True, type=_b)
parser.add_argument("--max_ner_type_len", type=int, default=80)
parser.add_argument("--max_ner_type_len_gt", type=int, default=20)
parser.add_argument("--freeze_clip", default=True, type=_b)
parser.add_argument("--prompt_mlp_type", type=str, default="clipcap")
parser.add_argument("--map_size", nargs="+", type=int)
parser.add_argument("--no_mapping", default=False, type=_b)
parser.add_argument("--mapping_loss_weight", type=float, default=1.0)
parser.add_argument("--img_size", type=int, default=768)
parser.add_argument("--only_image", default=False, type=_b)
parser.add_argument("--use_secla", default=True, type=_b)
parser.add_argument("--num_sentences", type=int, default=8)
parser.add_argument("--adapter_dim", type=int, default=96)
parser.add_argument("--project_name", type=str, default="news_cap")
parser.add_argument("--experiment_name", type=str, default="vacnic_mi355x")
parser.add_argument("--offline_wandb", default=True, type=_b)
parser.add_argument("--perturb", default=False, type=_b)
parser.add_argument("--no_clip_norm", default=True, type=_b)
parser.add_argument("--init_attn_weight", default=False, type=_b)
parser.add_argument("--margin", type=float, default=1.0)
parser.add_argument("--alpha", type=float, default=0.5)
# additions for the synthetic driver
parser.add_argument("--steps_per_epoch", type=int, default=20)
parser.add_argument("--log_every", type=int, default=5)
parser.add_argument("--val_steps", type=int, default=0, help="synthetic validation batches per epoch (eval_epoch, TRAIN:391-447); 0 = skip")
parser.add_argument("--test_steps", type=int, default=0, help="synthetic test batches generated after training (TRAIN:480-530); 0 = skip")
parser.add_argument("--launch_plan", default=True, type=_b, help="record the step's launch sequence once per batch shape and replay it from "
                    "C++ (vacnic_amd.training.PlannedTrainStep: one C call per step instead of ~1400 Python round trips); False = eager")
parser.add_argument("--resume", type=str, default="", help="checkpoint written by a previous run (<out_dir>/<experiment_name>last.pt): "
                    "weights, AdamW moments, LR-schedule position and dropout RNG are restored and the step count continues")

PLM = {"facebook/bart-base": dict(d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12,
                                  decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072),
       "facebook/bart-large": dict(), "patrickvonplaten/bart-large-fp32": dict()}
# from_pretrained(plm_type) (TRAIN:743) also inherits the checkpoint's dropout probabilities (attention / activation dropout 0.1 in
# both hub configs — BartConfig's own defaults are 0.0): vacnic_amd.config.HUB_MODEL_DROPOUTS
CLIP = {"ViT-B/32": dict(width=768, layers=12, patch_size=32, output_dim=512), "ViT-B/16": dict(width=768, layers=12, patch_size=16, output_dim=512),
        "ViT-L/14": dict(width=1024, layers=24, patch_size=14, output_dim=768)}


def run(args, batches=None):
    import torch
    import torch.distributed as dist
    from vacnic_amd import ops, streams, synthetic
    from vacnic_amd.config import HUB_MODEL_DROPOUTS, ClipVisionConfig, VacnicConfig
    from vacnic_amd.ddp import DistributedDataParallel
    from vacnic_amd.training import (FusedAdamW, PlannedTrainStep, TrainArgs, build_models, eval_epoch, gen_caption_from_loader_bart, to_device,
                                     train_step)

    if not args.no_clip_loss or not args.freeze_clip:
        raise NotImplementedError("CLIP contrastive loss / CLIP fine-tuning are out of scope (SURVEY §2 row 20): pass --no_clip_loss True --freeze_clip True")
    if not args.no_mapping and not args.use_secla and not args.only_image:
        raise ValueError("--use_secla False: the reference's pooled face-name branch (TRAIN:331-345) fails inside its own encoder "
                         "(add_ner_ffn=False -> attention-mask size ValueError); pass --use_secla True or --no_mapping True")
    local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        # torch.distributed is the control plane (rendezvous, the RCCL communicator id); the gradient all-reduce itself is RCCL
        # through the C-ABI on the reducer's native path (vacnic_amd/ddp.py).  VACNIC_DIST_BACKEND=nccl + VACNIC_DDP_COMM=wgrad: the
        # reference's arrangement (TRAIN:87, collectives through ProcessGroupNCCL)
        be = os.environ.get("VACNIC_DIST_BACKEND", "gloo" if os.environ.get("VACNIC_DDP_COMM", "native") == "native" else "nccl")
        dist.init_process_group(backend=be, **({"device_id": torch.device("cuda", local)} if be == "nccl" else {}))
    ops.Rng.manual_seed(int(args.seed) + rank)
    streams.enable(True)                     # guide forward + weight gradients on side streams
    vkw = CLIP[args.clip_type]
    cfg = VacnicConfig(enc_fusion_layer=list(args.enc_fusion_layer or []), dim_common=args.dim_common, prompt_size=args.prompt_size,
                       max_ner_type_len=args.max_ner_type_len, max_ner_type_len_gt=args.max_ner_type_len_gt,
                       only_image=args.only_image, clip_width=vkw["width"], prompt_mlp_type=args.prompt_mlp_type, map_size=args.map_size,
                       init_attn_weight=args.init_attn_weight,
                       **PLM[args.plm_type], **HUB_MODEL_DROPOUTS[args.plm_type]).validate()
    vcfg = ClipVisionConfig(**vkw)
    model, guide, _ = build_models(cfg, vcfg, device="cuda", seed=int(args.seed) % (2 ** 31), init="device")
    # Steps per epoch: the reference derives num_training_steps from the dataset (num_epoch * train_size / batch, TRAIN:99); with
    # a real shard an epoch is len(loader) steps, and that value — not the synthetic driver's --steps_per_epoch — must size the
    # linear schedule and the resume arithmetic (a schedule sized for 20 steps would sit at lr = 0 for the rest of the run).
    shard_loader = None
    steps_per_epoch = int(args.steps_per_epoch)
    if batches is None and args.data_type == "shard":
        from vacnic_amd import data
        shard_loader = data.PrefetchLoader(data.ShardReader(os.path.join(args.data_dir, "train.vshard")), args.train_batch_size, rank=rank,
                                           world=world, seed=int(args.seed) % 65536)
        steps_per_epoch = len(shard_loader)
        if steps_per_epoch <= 0:
            raise ValueError("the training shard holds fewer samples than one global batch")
    total_steps = int(args.num_epoch) * steps_per_epoch          # TRAIN:99 (not divided by world size there either)
    targs = TrainArgs(lr_bart=args.lr_bart, weight_decay=args.weight_decay, warmup_rate=args.warmup_rate,
                      num_training_steps=total_steps, margin=args.margin, alpha=args.alpha,
                      mapping_loss_weight=args.mapping_loss_weight, use_secla=args.use_secla, no_mapping=args.no_mapping,
                      no_clip_norm=args.no_clip_norm, clip_norm=args.clip_norm)
    net = DistributedDataParallel(model, device_ids=[local], output_device=local) if world > 1 else model
    opt = FusedAdamW(model.arena, lr=args.lr_bart, weight_decay=args.weight_decay, num_warmup_steps=args.warmup_rate * total_steps,
                     num_training_steps=total_steps, world_size=world)
    step, t0, hist = 0, time.time(), []
    plans = {}
    min_val_loss = 999.0                      # TRAIN:452
    start_step = 0
    if args.resume:
        from vacnic_amd import checkpoint
        start_step = int(checkpoint.load_checkpoint(args.resume, net, opt, rank=rank)["step"])
        step = start_step
    for epoch in range(start_step // steps_per_epoch, int(args.num_epoch)):
        if batches is not None:
            it = batches
        elif shard_loader is not None:
            # packed pre-tokenised shard (vacnic_amd/data.py): sampler + collate + pinned staging + async H2D in the loader
            it = shard_loader
            it.set_epoch(epoch)
        else:
            it = (synthetic.make_batch(cfg, args.train_batch_size, S=args.article_max_length, T=min(64, args.caption_max_length),
                                       seed=int(args.seed) % 65536, rank=rank, step=epoch * steps_per_epoch + i)
                  for i in range(steps_per_epoch))
        for bi, batch in enumerate(it):
            if epoch * steps_per_epoch + bi < start_step:
                continue                                            # already consumed before the checkpoint
            if batches is None and step >= total_steps:
                raise RuntimeError(f"step {step} beyond the schedule's {total_steps} training steps: the learning rate would be 0")
            ready = None
            if isinstance(batch, tuple):                            # (device batch, copy-stream event) from the PrefetchLoader
                batch, ready = batch
            else:
                batch = to_device(batch, "cuda")
            g_ = guide if not args.only_image else None
            if getattr(args, "launch_plan", True):
                # one recorded plan per batch geometry (a collated shard batch is padded to its own maxima): the first batch of a
                # geometry runs eagerly, the second is recorded while it runs, later ones are one replay each; at most 4 plans are kept
                # (each owns a private allocator pool)
                sig = tuple((k, tuple(v.shape)) for k, v in sorted(batch.items()))
                ent = plans.get(sig)
                if ent is None:
                    plans[sig] = "seen"
                    out4 = train_step(net, g_, opt, batch, targs, ready)
                elif ent == "seen" and len([v for v in plans.values() if v != "seen"]) < 4:
                    if ready is not None:
                        torch.cuda.current_stream().wait_event(ready)
                    ent = plans[sig] = PlannedTrainStep(net, g_, opt, targs, batch, warmup=0)       # (the recorded step IS this batch's step)
                    out4 = ent.out4
                elif ent == "seen":
                    out4 = train_step(net, g_, opt, batch, targs, ready)
                else:
                    out4 = ent(batch, ready)
            else:
                out4 = train_step(net, g_, opt, batch, targs, ready)
            step += 1
            if step % args.log_every == 0 and rank == 0:          # ONE device->host sync per log interval (the reference does 4 per step)
                tot, txt, secla, colam = out4.tolist()
                rec = {"step": step, "loss": tot, "text loss": txt, "face name loss": secla, "margin loss": colam,
                       "samples_per_s": round((step - start_step) * args.train_batch_size * world / (time.time() - t0), 2)}
                hist.append(rec)
                print(json.dumps(rec), flush=True)
        if args.val_steps > 0 and rank == 0:
            # TRAIN:455-470: validation pass per epoch; the best model and its teacher-forced outputs are kept
            vb = (synthetic.make_batch(cfg, args.val_batch_size, S=args.article_max_length, T=min(64, args.caption_max_length),
                                       seed=(int(args.seed) + 7919) % 65536, rank=0, step=i) for i in range(args.val_steps))
            val_loss, vdict = eval_epoch(net, vb)
            print(json.dumps({"epoch": epoch, "validation loss": val_loss}), flush=True)
            if val_loss < min_val_loss and args.out_dir:
                min_val_loss = val_loss
                os.makedirs(args.out_dir, exist_ok=True)
                from vacnic_amd import checkpoint
                checkpoint.save_checkpoint(os.path.join(args.out_dir, args.experiment_name + ".pt"), net, opt, step=step, rank=rank, config=dict(cfg.__dict__), vision=dict(vcfg.__dict__))
                with open(os.path.join(args.out_dir, args.experiment_name + "v.json"), "w") as f:
                    json.dump(vdict, f)
    if args.test_steps > 0 and rank == 0:
        # TRAIN:480-530,845-860: beam-search captions for the test split, written next to the checkpoints
        tb = (synthetic.make_batch(cfg, args.test_batch_size, S=args.article_max_length, T=min(64, args.caption_max_length),
                                   seed=(int(args.seed) + 104729) % 65536, rank=0, step=i) for i in range(args.test_steps))
        tdict = gen_caption_from_loader_bart(net, tb, args.beam_size, args.max_length, plm_type=args.plm_type)
        if args.out_dir:
            os.makedirs(args.out_dir, exist_ok=True)
            with open(os.path.join(args.out_dir, args.experiment_name + ".json"), "w") as f:
                json.dump(tdict, f)
        print(json.dumps({"test captions": len(tdict), "first": tdict[0]["gen"][0][:12] if tdict else []}), flush=True)
    torch.cuda.synchronize()
    if rank == 0 and args.out_dir:
        os.makedirs(args.out_dir, exist_ok=True)
        from vacnic_amd import checkpoint          # MFULL-named state_dict + optimizer/schedule/RNG (TRAIN:472 pickles the module object)
        checkpoint.save_checkpoint(os.path.join(args.out_dir, args.experiment_name + "last.pt"), net, opt, step=step, rank=rank, config=dict(cfg.__dict__), vision=dict(vcfg.__dict__))
    if world > 1:
        dist.destroy_process_group()
    return hist


if __name__ == "__main__":
    run(parser.parse_args())
