"""CPU, world_size 2 over gloo: the N>1 path of the data-parallel reducer — parameter broadcast at construction,
exact pending-write tracking, per-bucket launch as soon as a bucket is complete, SUM all-reduce, and AdamW's 1/world
averaging contract.  (The compute kernels need the GPU; here gradients are written into the arena by hand.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vacnic_amd import ddp
        from vacnic_amd.config import VacnicConfig
        from vacnic_amd.models.mmbart import BartForMultiModalGeneration
        torch.manual_seed(100 + rank)                      # ranks start with DIFFERENT weights
        cfg = VacnicConfig(d_model=768, encoder_layers=1, decoder_layers=1, encoder_attention_heads=12, decoder_attention_heads=12,
                           encoder_ffn_dim=64, decoder_ffn_dim=64, enc_fusion_layer=[0], dim_common=768, clip_width=768,
                           vocab_size=2048, max_position_embeddings=64)
        m = BartForMultiModalGeneration(cfg, enc_fusion_layer=[0], dim_common=768, prompt_size=2)
        m.finalize("cpu")
        w = ddp.DistributedDataParallel(m, bucket_bytes=1 << 20)
        # (i) after construction every rank holds rank 0's parameters
        ref = m.arena.flat32.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, m.arena.flat32), "ctor broadcast"
        assert w.module is m and w.world == 2
        tr = w.tracker
        assert ddp.TRACKER is tr and len(tr.buckets) > 3
        # (ii) forward registers pending writes; backward retires them; buckets launch exactly when complete
        params = [p for p in m.model.parameters()]
        for p in params:
            ddp.expect(True, p.grad)
        ddp.expect(True, m.model.shared.weight.grad)        # tied matrix: second writer (lm_head)
        assert sum(tr.pending.values()) >= len(params) + 1
        launched_before = set(w.launched)
        for p in reversed(params):
            p.grad.fill_(float(rank + 1))
            ddp.done(p.grad)
        assert len(w.launched) > len(launched_before), "buckets must launch during backward"
        shared_bucket = tr._bucket_of(m.model.shared.weight.grad)
        assert any(tr.pending[s] == 1 for s in shared_bucket), "bucket with the tied matrix still waits for its 2nd writer"
        assert not all(s in w.launched for s in shared_bucket)
        m.model.shared.weight.grad.add_(10.0 * (rank + 1))
        ddp.done(m.model.shared.weight.grad)
        w.reduce_gradients()
        # (iii) SUM over ranks everywhere (AdamW applies 1/world)
        for p in params:
            want = 3.0 + (30.0 if p is m.model.shared.weight else 0.0)
            assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (p.shape, p.grad.flatten()[:3])
        assert all(v == 0 for v in tr.pending.values()) and not w.works and not w.launched
        # unused parameters (never expected/done) are still reduced by reduce_gradients
        m.arena.grad.fill_(float(rank))
        w.reduce_gradients()
        assert torch.allclose(m.arena.grad, torch.ones_like(m.arena.grad))
        # bf16 gradient transport: buckets rounded to bf16, summed, widened back (within bf16 resolution of the fp32 sum)
        w16 = ddp.DistributedDataParallel(m, bucket_bytes=1 << 20, grad_transport="bf16")
        gen = torch.Generator().manual_seed(7 + rank)
        mine = torch.randn(m.arena.n, generator=gen)
        both = mine.clone()
        dist.all_reduce(both)
        m.arena.grad.copy_(mine)
        w16.reduce_gradients()
        err = (m.arena.grad - both).abs().max().item()
        assert err <= 2.0 ** -7 * both.abs().max().item() and err > 0.0, err
        assert not w16.works and not w16.launched
        # reduce_and_step: the optimizer is driven bucket by bucket, each range only after ITS all-reduce has finished, every
        # element exactly once, schedule first
        class FakeOpt:
            def __init__(self):
                self.calls, self.seen = [], torch.zeros(m.arena.n, dtype=torch.bool)

            def begin_step(self):
                self.calls.append("begin")

            def step_range(self, a, b):
                assert self.calls and self.calls[0] == "begin"
                assert torch.allclose(m.arena.grad[a:b], torch.full((b - a,), 3.0)), "range stepped before its bucket was reduced"
                assert not self.seen[a:b].any()
                self.seen[a:b] = True
                self.calls.append((a, b))

            def step(self, clip_norm=None):
                self.calls.append(("full", clip_norm))
        m.arena.grad.fill_(float(rank + 1))
        fo = FakeOpt()
        w.reduce_and_step(fo)
        assert fo.seen.all() and len(fo.calls) == 1 + len(tr.buckets) and not w.works and not w.launched and not w.order
        m.arena.grad.fill_(float(rank + 1))
        fo = FakeOpt()
        w.reduce_and_step(fo, clip_norm=0.1)                # clipping needs the global norm: plain reduce, one step
        assert fo.calls == [("full", 0.1)] and torch.allclose(m.arena.grad, torch.full_like(m.arena.grad, 3.0))
        # launch plans at world > 1 (training.PlannedTrainStep): while a step is RECORDED the reducer's host-side work goes through
        # ddp.HOST_HOOK — each piece runs at once and is kept as a host action at its position between the recorded commands
        # (here: the optimizer's range updates); a REPLAY re-runs the host actions and the commands in that order, with no
        # tracker activity at all.  The replay must reduce a fresh set of gradients exactly like an eager step does.
        plan = []

        class PlanOpt:                                      # stands for the C-ABI calls a plan records (AdamW per bucket range)
            def begin_step(self):
                plan.append(("cmd", "begin"))

            def step_range(self, a, b):
                plan.append(("cmd", (a, b)))

        ddp.HOST_HOOK = lambda fn: (plan.append(("host", fn)), fn())
        ddp.TRACKER = tr                                    # (the bf16-transport wrapper above installed its own)
        m.arena.grad.fill_(float(rank + 1))                 # (alignment gaps between the parameter slots included)
        for p in params:
            ddp.expect(True, p.grad)
        for p in reversed(params):
            p.grad.fill_(float(rank + 1))
            ddp.done(p.grad)                                # buckets complete during "backward": host actions recorded in between
        n_in_backward = sum(1 for k, _ in plan if k == "host")
        assert n_in_backward >= 1, "bucket launches during backward must be recorded as host actions"
        w.reduce_and_step(PlanOpt())
        ddp.HOST_HOOK = None
        kinds = [k for k, _ in plan]
        ranges = [v for k, v in plan if k == "cmd" and v != "begin"]
        assert len(ranges) == len(tr.buckets) and kinds.count("host") == 2 * len(tr.buckets) + 1, (len(ranges), kinds.count("host"))
        assert torch.allclose(m.arena.grad, torch.full_like(m.arena.grad, 3.0)) and not w.launched and not w.ready
        for i, (k, v) in enumerate(plan):                   # every range is preceded by a host action (the wait for ITS bucket)
            if k == "cmd" and v != "begin":
                assert plan[i - 1][0] == "host"
        for rep in range(2):                                # replays: other gradients, no tracker calls, only the recorded sequence
            gen = torch.Generator().manual_seed(50 + 10 * rep + rank)
            mine = torch.randn(m.arena.n, generator=gen)
            both = mine.clone()
            dist.all_reduce(both)
            m.arena.grad.copy_(mine)
            stepped = torch.zeros(m.arena.n, dtype=torch.bool)
            for k, v in plan:
                if k == "host":
                    v()
                elif v != "begin":
                    a, b = v
                    assert torch.equal(m.arena.grad[a:b], both[a:b]), "a replayed range ran before its bucket's all-reduce"
                    stepped[a:b] = True
            assert stepped.all() and torch.equal(m.arena.grad, both)
            assert not w.works and not w.launched and not w.ready and all(v == 0 for v in tr.pending.values())
        # expect() is ignored when the caller says no grad is needed (inference under no_grad)
        ddp.expect(False, params[0].grad)
        assert all(v == 0 for v in tr.pending.values())
        q.put((rank, "ok"))
    except Exception as e:          # surface the failure in the parent
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_ddp_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
