"""MI355X, world_size 2 on ONE GPU over gloo: the data-parallel reducer driven by the real HIP training step — tracker,
bucket launches on the comm stream while backward is still running, waits on the weight-gradient / branch streams, SUM
all-reduce, fp32 and bf16 transport — and the SURVEY §8d / §8e gate: N ranks x B samples must equal one rank evaluated
per B-sample shard and averaged (SECLA keeps per-rank negatives, TRAIN:326-330), and for the sample-separable terms
(CE with equal token counts, CoLaM) also one rank on the concatenated 2B batch.

RCCL itself needs one GPU per rank (the driver's 8-GPU node); everything above the collective call is exercised here."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, S, T, F = 3, 48, 12, 3


def _cfgs():
    from vacnic_amd.config import ClipVisionConfig, VacnicConfig
    cfg = VacnicConfig(d_model=768, encoder_layers=2, decoder_layers=2, encoder_attention_heads=12, decoder_attention_heads=12,
                       encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=128, dropout=0.0)
    vcfg = ClipVisionConfig(width=128, layers=1, patch_size=16, image_size=32, output_dim=64)
    return cfg, vcfg


def _batch(cfg, lo, hi):
    """samples [lo, hi) of the global batch of 2B samples (full-length captions: equal CE token counts per shard)."""
    from vacnic_amd import synthetic
    full = synthetic.make_batch(cfg, 2 * B, S=S, T=T, F=F, seed=11, image_size=32, full_length=True)
    return {k: v[lo:hi].contiguous().cuda() for k, v in full.items()}


def _fwd_bwd(model, guide, batch, args):
    from vacnic_amd import streams
    from vacnic_amd.ddp import DistributedDataParallel
    from vacnic_amd.training import forward_losses
    net = model.module if isinstance(model, DistributedDataParallel) else model
    net.arena.grad.zero_()
    total, out4, _ = forward_losses(model, guide, batch, args)
    with torch.autograd.set_multithreading_enabled(False):
        total.backward()
    streams.join_all()
    if isinstance(model, DistributedDataParallel):
        model.reduce_gradients()
    torch.cuda.synchronize()
    return out4.tolist(), net.arena.grad.clone()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="2")      # two processes on ONE card: 2 x 2 hardware queues (vacnic_amd/__init__.py)
    try:
        torch.cuda.set_device(0)
        from vacnic_amd import ddp, streams
        from vacnic_amd.training import TrainArgs, build_models
        streams.enable(True)                                   # wgrad / branch side streams: the reducer must wait for them
        cfg, vcfg = _cfgs()
        model, guide, _ = build_models(cfg, vcfg, init="synthetic", seed=3)   # frozen guide / ViT: same "pretrained" weights on all ranks
        if rank != 0:                                          # the trainable network starts DIFFERENT: the ctor broadcast must fix it
            model.arena.flat32.mul_(1.0 + 0.01 * rank)
            model.arena.refresh_shadow()
        model.train()
        args = TrainArgs()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        res, kept = {}, {}
        for transport in ("fp32", "bf16"):
            w = ddp.DistributedDataParallel(model, bucket_bytes=8 << 20, grad_transport=transport)
            assert ddp.TRACKER is w.tracker and len(w.tracker.buckets) > 8
            launched_in_backward = []
            orig = w._launch_bucket

            def spy(start, orig=orig, log=launched_in_backward):       # tracker callback = a bucket completed DURING backward
                log.append(start)
                orig(start)
            w.tracker.on_ready = spy
            losses, gsum = _fwd_bwd(w, guide, _batch(cfg, rank * B, (rank + 1) * B), args)
            assert len(launched_in_backward) >= len(w.tracker.buckets) // 2, "buckets must launch while backward is running"
            gavg = gsum / world
            res[transport] = {"losses": losses, "probe": gavg[::9973].double().cpu().tolist(), "norm": gavg.double().norm().item()}
            kept[transport] = gavg
            ddp.TRACKER = None
        # the pipelined optimizer (reduce_and_step: AdamW bucket by bucket behind the all-reduces) == reduce, then one AdamW launch.
        # (i) on INJECTED gradients (a different one per rank) the two must agree bit for bit — the all-reduce and AdamW are both
        # order-independent per element; (ii) through the whole train_step against forward/backward + reduce + step: two separate
        # passes.  Round 3 had to gate (ii) at 5e-2: the LM-head dgrad summed its vocabulary chunks with split-K fp32 atomics whose
        # order varied when two processes shared the card, one bf16 element of dH rounded the other way about every second pass,
        # and AdamW's first step (sign(g) lr) turned that into 1-2 % of the update's norm.  The dgrad now sums through the GEMM's
        # ordered fix-up (ops.LMHEAD_FIXUP): what is left are the LayerNorm-parameter atomics (1e-7), and the gate is 1e-4 again.
        from vacnic_amd.training import FusedAdamW, train_step
        w = ddp.DistributedDataParallel(model, bucket_bytes=8 << 20)
        opt = FusedAdamW(model.arena, lr=1e-3, num_warmup_steps=0, num_training_steps=10, world_size=world)
        snap = [t.clone() for t in (model.arena.flat32, model.arena.exp_avg, model.arena.exp_avg_sq, opt.hyper)]
        mine = _batch(cfg, rank * B, (rank + 1) * B)
        state = (model.arena.flat32, model.arena.exp_avg, model.arena.exp_avg_sq, opt.hyper)

        def restore():
            for t, s_ in zip(state, snap):
                t.copy_(s_)
            model.arena.refresh_shadow(); model.arena.grad.zero_()

        g_inject = kept["fp32"] * (1.0 + 0.37 * rank) + 1e-7 * rank
        exact = []
        for pipelined in (True, False):
            restore()
            model.arena.grad.copy_(g_inject)
            if pipelined:
                w.reduce_and_step(opt)
            else:
                w.reduce_gradients()
                opt.step()
            torch.cuda.synchronize()
            exact.append([t.clone() for t in (model.arena.flat32, model.arena.flat16, model.arena.exp_avg, model.arena.exp_avg_sq, model.arena.grad)])
        res["pipelined_exact"] = [bool(torch.equal(a_, b_)) for a_, b_ in zip(*exact)]
        res["pipelined_exact_moved"] = bool((exact[0][0] != snap[0]).any().item())
        results = []
        for pipelined in (True, False):
            restore()
            if pipelined:
                train_step(w, guide, opt, mine, args)                          # -> reduce_and_step
            else:
                from vacnic_amd.training import forward_losses
                total, _, _ = forward_losses(w, guide, mine, args)
                with torch.autograd.set_multithreading_enabled(False):
                    total.backward()
                streams.join_all()
                w.reduce_gradients()
                opt.step()
            torch.cuda.synchronize()
            results.append((model.arena.flat32.clone(), model.arena.flat16.float().clone(), model.arena.grad.abs().max().item()))
        res["pipelined_update_rel_diff"] = ((results[0][0] - results[1][0]).norm() / (results[1][0] - snap[0]).norm()).item()
        res["pipelined_grad_left"] = results[0][2]
        res["pipelined_shadow_ok"] = bool(torch.equal(results[0][1] != 0, results[1][1] != 0))
        ddp.TRACKER = None
        for t, s_ in zip((model.arena.flat32, model.arena.exp_avg, model.arena.exp_avg_sq), snap):
            t.copy_(s_)
        model.arena.refresh_shadow(); model.arena.grad.zero_()
        # the single-process references, computed by rank 0 with the (broadcast) rank-0 weights and no reducer
        if rank == 0:
            shard = [_fwd_bwd(model, guide, _batch(cfg, s * B, (s + 1) * B), args) for s in range(world)]
            whole = _fwd_bwd(model, guide, _batch(cfg, 0, world * B), args)
            g_ref = sum(g for _, g in shard) / world               # mean over shards of the per-shard gradient
            res["shard_losses"] = [l for l, _ in shard]
            res["whole_losses"] = whole[0]
            res["ref_probe"] = g_ref[::9973].double().cpu().tolist()
            for transport, g in kept.items():
                res[transport]["rel_vs_shard_mean"] = ((g - g_ref).double().norm() / g_ref.double().norm()).item()
        q.put((rank, "ok", res))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_ranks_match_one_rank_per_shard_and_on_the_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    by_rank = {r[0]: r[2] for r in res}
    ref = by_rank[0]
    for transport, tol in (("fp32", 1e-3), ("bf16", 6e-3)):           # bf16 transport rounds every bucket twice (2^-9 each)
        for r in (0, 1):
            # every loss term of rank r equals the single-process evaluation of shard r (total, txt, secla, colam)
            for got, want in zip(by_rank[r][transport]["losses"], ref["shard_losses"][r]):
                assert abs(got - want) <= 1e-3 * abs(want), (transport, r, by_rank[r][transport]["losses"], ref["shard_losses"][r])
        # averaged gradient of the 2-rank step == mean over shards of the single-process gradient
        assert ref[transport]["rel_vs_shard_mean"] <= tol, (transport, ref[transport]["rel_vs_shard_mean"])
        # both ranks hold the same reduced gradient (probe of every 9973rd element + norm)
        assert by_rank[0][transport]["probe"] == by_rank[1][transport]["probe"], transport   # plain lists: no tensors through the queue
        assert by_rank[0][transport]["norm"] == by_rank[1][transport]["norm"], transport
    # AdamW pipelined bucket by bucket behind the all-reduces == one AdamW launch after the reduce: bit for bit on injected gradients
    # (weights, bf16 shadow, both moments, cleared gradient arena), and through two separate train_step passes to 1e-4 (see _worker)
    for r in (0, 1):
        assert all(by_rank[r]["pipelined_exact"]) and by_rank[r]["pipelined_exact_moved"], by_rank[r]["pipelined_exact"]
        assert by_rank[r]["pipelined_update_rel_diff"] <= 1e-4, by_rank[r]["pipelined_update_rel_diff"]
        assert by_rank[r]["pipelined_grad_left"] == 0.0 and by_rank[r]["pipelined_shadow_ok"]
    # concatenated batch on one rank: CE (equal token counts per shard) and CoLaM are sample means -> equal to the shard mean;
    # SECLA is NOT (in-batch negatives are per rank by design, TRAIN:326-330): the N-rank value is the per-shard mean
    whole = ref["whole_losses"]
    mean_shard = [sum(l[i] for l in ref["shard_losses"]) / 2.0 for i in range(4)]
    assert abs(whole[1] - mean_shard[1]) <= 1e-3 * abs(mean_shard[1]), ("txt", whole, mean_shard)
    assert abs(whole[3] - mean_shard[3]) <= 1e-3 * abs(mean_shard[3]) + 1e-5, ("colam", whole, mean_shard)
    ddp_secla = (by_rank[0]["fp32"]["losses"][2] + by_rank[1]["fp32"]["losses"][2]) / 2.0
    assert abs(ddp_secla - mean_shard[2]) <= 1e-3 * abs(mean_shard[2]), ("secla per-shard mean", ddp_secla, mean_shard[2])


def _worker_plan(rank, world, port, q):
    """2 ranks on one GPU: the launch plan at world > 1 (the reducer's collectives as host actions at the plan's marks)
    must train like the eager DDP step — same losses over three batches, same final weights."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="2")      # two processes on ONE card: 2 x 2 hardware queues (vacnic_amd/__init__.py)
    try:
        torch.cuda.set_device(0)
        from vacnic_amd import _lib, ddp, ops, streams, synthetic
        from vacnic_amd.training import FusedAdamW, PlannedTrainStep, TrainArgs, build_models, train_step
        streams.enable(True)
        cfg, vcfg = _cfgs()
        args = TrainArgs(num_training_steps=20, warmup_rate=0.1, lr_bart=1e-4)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        batches = []
        for i in range(3):
            full = synthetic.make_batch(cfg, 2 * B, S=S, T=T, F=F, seed=60 + i, image_size=32)
            batches.append({k: v[rank * B:(rank + 1) * B].contiguous().cuda() for k, v in full.items()})
        res = {}
        for transport in ("fp32", "bf16"):
            runs = {}
            for planned in (False, True):
                ops.Rng.manual_seed(3); ops.Rng.device_counter().zero_()
                model, guide, _ = build_models(cfg, vcfg, init="synthetic", seed=3)
                model.train()
                w = ddp.DistributedDataParallel(model, bucket_bytes=8 << 20, grad_transport=transport)
                opt = FusedAdamW(model.arena, lr=args.lr_bart, weight_decay=args.weight_decay, num_warmup_steps=2, num_training_steps=20,
                                 world_size=world)
                order = batches[1:] + batches[:1]
                if planned:
                    step = PlannedTrainStep(w, guide, opt, args, batches[0], warmup=2)          # 2 eager steps + the recorded (executed) one
                    hosts = [m for m in step.marks if callable(m[1])]
                    nb = len(w.tracker.buckets)
                    assert len(hosts) == 2 * nb + 1, (len(hosts), nb)                          # launch + wait per bucket, one reset
                    # host marks in order: nb bucket launches, then {wait, AdamW range} per bucket, then the reset.  Launches issued by
                    # reduce_and_step itself sit right in front of the schedule kernel that precedes the first wait; the others
                    # were issued by the tracker while backward was still being recorded
                    first_wait = hosts[nb][0]
                    assert sum(1 for idx, _ in hosts[:nb] if idx < first_wait - 1) >= nb // 2, "bucket launches must sit inside the backward"
                    losses = []
                    for b in order:
                        c0 = _lib.CALLS
                        losses.append(step(b).tolist())
                        calls = _lib.CALLS - c0
                        assert calls <= len(step.marks) + 1 + (2 * nb if transport == "bf16" else 0), calls   # one replay call per plan segment (+ the casts of bf16 transport)
                    step.close()
                else:
                    for _ in range(3):
                        train_step(w, guide, opt, batches[0], args)
                    losses = [train_step(w, guide, opt, b, args).tolist() for b in order]
                torch.cuda.synchronize()
                runs[planned] = (losses, model.arena.flat32.clone())
                ddp.TRACKER = None
                del w, opt, model, guide
            dl = max(abs(a - b) / max(abs(b), 1e-6) for la, lb in zip(runs[True][0], runs[False][0]) for a, b in zip(la, lb))
            dw = ((runs[True][1] - runs[False][1]).double().norm() / runs[False][1].double().norm()).item()
            res[transport] = {"loss_rel": dl, "weight_rel": dw, "finite": all(abs(v) < 1e30 for l in runs[True][0] for v in l)}
        q.put((rank, "ok", res))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_planned_step_at_world_2_trains_like_the_eager_ddp_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_plan, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    for r in res:
        for transport in ("fp32", "bf16"):
            v = r[2][transport]
            assert v["finite"] and v["loss_rel"] <= 2e-3 and v["weight_rel"] <= 1e-4, (r[0], transport, v)


def _worker_rccl(q, port):
    """ONE rank, the REAL communication library: a one-rank RCCL communicator on the GPU and the whole reducer path forced on
    (DistributedDataParallel(force_reducer=True)) — tracker, bucket all-reduces launched on the comm stream while backward runs,
    waits, pipelined AdamW, eager and as host actions of a launch plan, fp32 and bf16 transport.  A sum over one rank is the
    identity, so the run must train exactly like the plain single-GPU step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        from vacnic_amd import ddp, ops, streams, synthetic
        from vacnic_amd.training import FusedAdamW, PlannedTrainStep, TrainArgs, build_models, train_step
        streams.enable(True)
        cfg, vcfg = _cfgs()
        args = TrainArgs(num_training_steps=20, warmup_rate=0.1, lr_bart=1e-4)
        dist.init_process_group("gloo", rank=0, world_size=1)          # control plane only: the data plane is RCCL through the C-ABI
        batches = [{k: v.cuda() for k, v in synthetic.make_batch(cfg, B, S=S, T=T, F=F, seed=70 + i, image_size=32).items()} for i in range(3)]
        order = batches[1:] + batches[:1]
        runs = {}
        for mode in ("plain", "rccl_eager", "rccl_plan", "rccl_plan_bf16", "torch_plan"):
            ops.Rng.manual_seed(3); ops.Rng.device_counter().zero_()
            model, guide, _ = build_models(cfg, vcfg, init="synthetic", seed=3)
            model.train()
            if mode == "torch_plan":                       # the torch.distributed data plane (collectives as host actions of the plan)
                ddp._COMM_MODE = "wgrad"
            net = model if mode == "plain" else ddp.DistributedDataParallel(model, bucket_bytes=8 << 20, force_reducer=True,
                                                                           grad_transport="bf16" if mode.endswith("bf16") else "fp32")
            ddp._COMM_MODE = "native"
            if mode == "torch_plan":
                assert net.native is None and net.comm_on_wgrad
            elif mode != "plain":
                assert net.active and net.world == 1 and ddp.TRACKER is net.tracker and net.native is not None, "RCCL through the C-ABI (vacnic_allreduce_bucket) must be the data plane"
            opt = FusedAdamW(model.arena, lr=args.lr_bart, weight_decay=args.weight_decay, num_warmup_steps=2, num_training_steps=20)
            if "plan" in mode:
                step = PlannedTrainStep(net, guide, opt, args, batches[0], warmup=2)
                hosts = sum(1 for m in step.marks if callable(m[1]))
                assert hosts == (2 * len(net.tracker.buckets) + 1 if mode == "torch_plan" else 0), (mode, hosts)   # native: plan commands, no host actions
                losses = [step(b).tolist() for b in order]
                step.close()
            else:
                for _ in range(3):
                    train_step(net, guide, opt, batches[0], args)
                losses = [train_step(net, guide, opt, b, args).tolist() for b in order]
            torch.cuda.synchronize()
            runs[mode] = (losses, model.arena.flat32.clone())
            ddp.TRACKER = None
            del net, opt, model, guide
        res = {}
        for mode in ("rccl_eager", "rccl_plan", "rccl_plan_bf16", "torch_plan"):
            dl = max(abs(a - b) / max(abs(b), 1e-6) for la, lb in zip(runs[mode][0], runs["plain"][0]) for a, b in zip(la, lb))
            dw = ((runs[mode][1] - runs["plain"][1]).double().norm() / runs["plain"][1].double().norm()).item()
            res[mode] = (dl, dw)
        q.put(("ok", res))
    except Exception:
        import traceback
        q.put(("FAIL: " + traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_one_rank_rccl_reducer_trains_like_the_plain_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(q, 30100 + (os.getpid() % 2000)))
    p.start()
    status, res = q.get(timeout=600)
    p.join(timeout=120)
    assert status == "ok", status
    for mode in ("rccl_eager", "rccl_plan", "torch_plan"):        # the identity all-reduce: the same training run (LayerNorm atomics aside)
        assert res[mode][0] <= 2e-3 and res[mode][1] <= 1e-4, (mode, res[mode])
    assert res["rccl_plan_bf16"][0] <= 5e-3 and res["rccl_plan_bf16"][1] <= 5e-4, res["rccl_plan_bf16"]    # gradients rounded to bf16 on the wire
