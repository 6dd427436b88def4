"""Entry point with the file name and flags of the reference's stand-alone caption generator (DDPINF:1-40,1017-1296) on the
MI355X-native decode path: load a trained model, one process per GPU, beam-search captions for the test split
(`--beam_size 5 --max_length 50 --length_penalty 2.0` are the settings of the published numbers, README.md:8), write
`<out_dir>/<model_name>beam{b}_max{m}t_S{seed}_lp{lp}.json` (DDPINF:1293).

    torchrun --nproc_per_node=N utils/test_mmbart_clip_ddp.py --model_dir OUT --model_name NAME --beam_size 5 --max_length 50 \
        --length_penalty 2.0 --test_batch_size 8 --data_type synthetic --test_steps 16

What differs from the reference, on purpose:
* the reference unpickles a whole module (`torch.load`, DDPINF:1085) and every rank decodes the WHOLE test set behind a DDP
  wrapper that is never used for a collective; here the checkpoint is tensors keyed by the reference's parameter names plus the
  model geometry (vacnic_amd/checkpoint.py), and the test batches are strided over the ranks (batch i -> rank i % world) with one
  object gather at the end — captions/s scales with the GPU count;
* batched decoding (`--test_batch_size > 1`), which the reference's loop cannot do (it indexes `[0]` of every batch);
* tokenizers and BLEU/ROUGE/CIDEr/METEOR scoring are outside SURVEY §8: the JSON holds token ids ("gt", "gen").
The frozen CLIP tower is loaded separately from the checkpoint, as at DDPINF:1091 (`clip.load`): with no pretrained weights in
this environment it is re-created from `--seed`, so pass the seed the trainer ran with.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_b = lambda x: (str(x).lower() == "true")    # noqa: E731
parser = argparse.ArgumentParser()
parser.add_argument("--local_rank", type=int, default=-1)
parser.add_argument("--seed", type=str, default="684331")
parser.add_argument("--gpu_ids", type=str, default="0")
parser.add_argument("--num_workers", type=int, default=16)
parser.add_argument("--article_max_length", type=int, default=512)
parser.add_argument("--plm_type", type=str, default="facebook/bart-base")
parser.add_argument("--clip_type", type=str, default="ViT-B/32")
parser.add_argument("--ent_start_token", type=str, default="no")
parser.add_argument("--ent_end_token", type=str, default="no")
parser.add_argument("--enc_fusion_layer", nargs="+", type=int)
parser.add_argument("--dec_fusion_layer", nargs="+", type=int)
parser.add_argument("--use_img_trans", default=False, type=_b)
parser.add_argument("--use_forget_gate", default=False, type=_b)
parser.add_argument("--cross_attn_type", type=int, default=5)
parser.add_argument("--dim_common", type=int, default=768)
parser.add_argument("--n_attn_heads", type=int, default=12)
parser.add_argument("--test_batch_size", type=int, default=1)
parser.add_argument("--beam_size", type=int, default=5)
parser.add_argument("--max_length", type=int, default=100)
parser.add_argument("--data_type", type=str, default="synthetic")
parser.add_argument("--data_dir", type=str, default="DATADIR")
parser.add_argument("--prompt_size", type=int, default=8)
parser.add_argument("--model_name", type=str, default="MODELNAME")
parser.add_argument("--model_dir", type=str, default="MODELDIR")
parser.add_argument("--num_sentences", type=int, default=8)
parser.add_argument("--dict_type", type=str, default="tune")
parser.add_argument("--length_penalty", type=float, default=1)
# additions
parser.add_argument("--out_dir", type=str, default="", help="where the caption JSON goes (default: --model_dir; the reference hard-codes OUTPUTDIR)")
parser.add_argument("--test_steps", type=int, default=8, help="--data_type synthetic: number of test batches")
parser.add_argument("--caption_max_length", type=int, default=100)


def run(args):
    import torch
    import torch.distributed as dist
    from vacnic_amd import checkpoint, synthetic
    from vacnic_amd.config import ClipVisionConfig, VacnicConfig
    from vacnic_amd.training import build_models, gen_caption_from_loader_bart

    local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(backend="nccl")                          # DDPINF:1078-1079
        dist.barrier()
    path = os.path.join(args.model_dir, args.model_name + ".pt")
    ck = torch.load(path, map_location="cpu", weights_only=False)       # DDPINF:1085
    meta = ck.get("meta", {})
    if "config" not in meta or "vision" not in meta:
        raise ValueError(f"{path} carries no model geometry (meta.config / meta.vision): write it with the trainer entry points of this repo")
    cfg = VacnicConfig(**meta["config"]).validate()
    vcfg = ClipVisionConfig(**meta["vision"])
    model, _, _ = build_models(cfg, vcfg, device="cuda", seed=int(args.seed) % (2 ** 31), init="device", with_guide=False)
    checkpoint.load_checkpoint(ck, model)
    model.eval()
    steps = list(range(rank, args.test_steps, world))
    if args.data_type == "shard":
        from vacnic_amd import data
        loader = data.PrefetchLoader(data.ShardReader(os.path.join(args.data_dir, "test.vshard")), args.test_batch_size, rank=rank, world=world,
                                     shuffle=False, drop_last=False, flip=False)

        def staged():
            for b, ev in loader:                      # (device batch, copy-stream event): decode only after the batch has landed
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                yield b
        batches = staged()
        steps = None
    else:
        batches = (synthetic.make_batch(cfg, args.test_batch_size, S=args.article_max_length, T=min(64, args.caption_max_length), seed=(int(args.seed) + 104729) % 65536,
                                        rank=0, step=i) for i in steps)
    torch.cuda.synchronize()
    t0 = time.time()
    local_out = gen_caption_from_loader_bart(model, batches, args.beam_size, args.max_length, length_penalty=args.length_penalty, plm_type=args.plm_type)
    torch.cuda.synchronize()
    dt = time.time() - t0
    keyed = {(steps[k] if steps is not None else k * world + rank): v for k, v in local_out.items()}
    parts = [keyed]
    if world > 1:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(keyed, parts, dst=0)
    if rank == 0:
        merged = {}
        for p in parts:
            merged.update(p)
        merged = {str(k): merged[k] for k in sorted(merged)}
        n_cap = sum(len(v["gen"]) for v in merged.values())
        out_dir = args.out_dir or args.model_dir
        os.makedirs(out_dir, exist_ok=True)
        tag = f"beam{args.beam_size}_max{args.max_length}t_S{args.seed}_lp{args.length_penalty}"
        with open(os.path.join(out_dir, args.model_name + tag + ".json"), "w") as f:          # DDPINF:1293
            json.dump(merged, f)
        print(json.dumps({"tag": tag, "captions": n_cap, "batches": len(merged), "n_gpus": world,
                          "captions_per_s": round(n_cap / dt, 2) if world == 1 else None, "rank0_seconds": round(dt, 3)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(parser.parse_args())
