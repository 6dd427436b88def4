"""Data-parallel gradient reduction over RCCL/xGMI — replaces torch.nn.parallel.DistributedDataParallel
(TRAIN:87, TRAINV:84, DDPINF:1089) for the arena-based model.

One process per GPU.  The path is pure data parallel: the only collectives are a parameter
broadcast at construction and one SUM all-reduce of the gradient arena per step; averaging (1/world) is
folded into the fused AdamW kernel (grad_scale).  Because gradients already live in ONE flat buffer there
is no flatten/unflatten: buckets are contiguous slices.

Data plane (default, `_COMM_MODE` "native"): RCCL through the C-ABI (csrc/comm.hip) on the weight-gradient
stream; torch.distributed (gloo) is the control plane only — rendezvous, the 128-byte communicator id, the
check that no two ranks share a GPU.  `VACNIC_DDP_COMM=wgrad|own` keeps the torch.distributed ("nccl" ==
RCCL on ROCm) data plane of rounds 1-3 for comparison; on CPU (gloo tests) the torch path is the only one.

Overlap with backward is exact, not heuristic: every op that will accumulate into a gradient slice
registers a pending write in forward (GradTracker.expect) and retires it in backward right after it
has enqueued its wgrad kernel (GradTracker.done).  When the pending count of a bucket reaches zero
the reducer fences the bucket's writer streams into the weight-gradient stream and launches that
bucket's all-reduce there, followed by a named event the compute stream waits for right before the
bucket's AdamW range — large buckets (default 256 MiB: xGMI is point-to-point, 7 links per GPU, so few
large collectives beat many 25 MiB ones) in the order backward completes them.
"""
import os

import torch
import torch.distributed as dist

_SKIP_COLLECTIVE = os.environ.get("VACNIC_DDP_SKIP_COLLECTIVE") == "1"
# Which HIP stream hands a bucket to the communication library:
#   "wgrad" (default when side streams are on): the weight-gradient stream — the stream most of a bucket's writers run on — waits
#           for the other writers and issues the collective; the compute stream picks the result up by waiting on the library's
#           work handle right before that bucket's AdamW range.  No stream of our own is added.
#   "own":  a dedicated comm stream (rounds 1-3).
# Why it matters: the step runs on FOUR streams (compute, weight gradients, frozen towers, small-token branches) and the GPU
# exposes four hardware queues to a process by default.  A fifth and sixth stream (ours + the library's internal one) make HIP
# multiplex streams onto queues, and two of the step's own streams then serialise: measured on one box with a ONE-rank RCCL
# communicator — no bytes on any link — 64.4 ms/step without the reducer, 69.3 with the dedicated comm stream, and worse with
# more queues (GPU_MAX_HW_QUEUES=6 / 8: 96 / 78 ms).  profiles/r4_ddp_one_rank_rccl.txt
#   "native" (default on GPUs, one rank per device): RCCL through the C-ABI (csrc/comm.hip) — the collective is a stream-ordered
#           launch on the weight-gradient stream, recorded into a launch plan like any kernel; torch.distributed only carries the
#           128-byte communicator id (over a gloo group: no GPU stream).  Four streams, no host action per bucket.
_COMM_MODE = os.environ.get("VACNIC_DDP_COMM", "native")
_EV_BASE = 256            # named-event slots [256, 512) of the library belong to the reducer (one per bucket)


def _share_a_gpu(pg):
    """do two ranks of the group sit on one card (CPU-side rehearsals of the multi-rank path on a one-GPU box)?"""
    import socket
    me = (socket.gethostname(), int(torch.cuda.current_device()))
    everyone = [None] * dist.get_world_size(pg)
    dist.all_gather_object(everyone, me, group=pg)
    return len(set(everyone)) != len(everyone)


class NativeComm:
    """one RCCL communicator through the C-ABI.  Raises when the ranks of the group do not sit on distinct GPUs (several ranks
    rehearsing on one card: RCCL refuses that) or librccl cannot be resolved: the caller then keeps the torch.distributed path."""

    def __init__(self, pg=None):
        from . import kernels as K
        rank, world = dist.get_rank(pg), dist.get_world_size(pg)
        ctl = pg
        if dist.get_backend(pg) != "gloo":                    # the id must not travel through a GPU collective (that would bring
            ctl = dist.new_group(backend="gloo")              # torch's own NCCL communicator and its stream to life)
        import socket
        me = (socket.gethostname(), int(torch.cuda.current_device()))
        everyone = [None] * world
        dist.all_gather_object(everyone, me, group=ctl)
        if len(set(everyone)) != world:
            raise RuntimeError("several ranks share a GPU")
        K.comm_load()
        obj = [K.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0, group=ctl)
        self.h = K.comm_init(obj[0], rank, world)
        self.rank, self.world = rank, world


class GradTracker:
    """Per-bucket count of gradient writes still outstanding in the current backward."""

    def __init__(self, arena, bucket_bytes=256 << 20):
        self.arena = arena
        self.base = arena.grad.data_ptr()
        self.buckets = arena.bucket_slices(bucket_bytes)        # reverse layout order
        self.starts = sorted(s for s, _ in self.buckets)
        self.pending = {s: 0 for s in self.starts}
        self.on_ready = None
        self.in_backward = False
        self._memo = {}

    def _bucket_of(self, t):
        """start offsets of the buckets a gradient view touches.  Gradient views are persistent slices of the arena, so the
        answer is memoised per (address, length): after the first step expect() / done() cost one dict lookup each."""
        key = (t.data_ptr(), t.numel())
        hit = self._memo.get(key)
        if hit is None:
            import bisect
            off = (key[0] - self.base) // 4
            if off < 0 or off + key[1] > self.arena.n:
                raise ValueError("GradTracker: tensor is not a view of the gradient arena")
            lo = bisect.bisect_right(self.starts, off) - 1
            hi = bisect.bisect_right(self.starts, off + max(key[1], 1) - 1) - 1
            hit = self._memo[key] = tuple(self.starts[lo:hi + 1])
        return hit

    def expect(self, t):
        if t is None:
            return
        for s in self._bucket_of(t):
            self.pending[s] += 1

    def done(self, t):
        if t is None:
            return
        for s in self._bucket_of(t):
            self.pending[s] -= 1
            if self.pending[s] == 0 and self.on_ready is not None:
                self.on_ready(s)

    def reset(self):
        for s in self.pending:
            self.pending[s] = 0


TRACKER = None        # set by DistributedDataParallel when world_size > 1; ops.py consults it
HOST_HOOK = None      # set by training.PlannedTrainStep while it records a launch plan: callable(fn) runs fn now AND at the same point
                      # of every replay (the plan is split there).  The reducer's host-side work — waiting for the side streams,
                      # launching a bucket's collective, making the compute stream wait for it — goes through _host().


def _host(fn):
    if HOST_HOOK is not None:
        HOST_HOOK(fn)
    else:
        fn()


def expect(need_grad, *tensors):
    """called from Function.forward (grad mode is off there, so the caller passes any(ctx.needs_input_grad))."""
    if TRACKER is not None and need_grad:
        for t in tensors:
            TRACKER.expect(t)


def done(*tensors):
    if TRACKER is not None:
        for t in tensors:
            TRACKER.done(t)


class DistributedDataParallel(torch.nn.Module):
    """DDP-shaped wrapper: `.module`, forward passthrough; call `reduce_gradients()` after backward()
    (or use `vacnic_amd.training.train_step`, which does).  With world_size 1 it is a no-op shell."""

    def __init__(self, module, device_ids=None, output_device=None, process_group=None, bucket_bytes=256 << 20,
                 overlap=True, grad_transport="fp32", force_reducer=False):
        """force_reducer: run the whole reducer (tracker, bucket launches on the comm stream, waits, pipelined AdamW) even on a
        ONE-rank process group — every collective then executes on the real communicator (RCCL on one GPU) although it has nobody
        to talk to.  A rehearsal aid for boxes with a single GPU (bench.py VACNIC_BENCH_FORCE_DDP=1, tests/test_ddp_gpu.py)."""
        super().__init__()
        global TRACKER
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force_reducer) and dist.is_available() and dist.is_initialized())
        self.arena = module.arena
        if self.arena is None or self.arena.grad is None:
            raise RuntimeError("wrap a finalized trainable model (module.finalize(device) first)")
        self.works = []
        self.launched = set()
        self.order = []                   # bucket starts in launch order
        self.ready = {}                   # bucket start -> completion handle (event on the comm stream / index into works)
        self.comm_stream = None
        self.comm_on_wgrad = False
        self.native = None                # NativeComm when the data plane is RCCL through the C-ABI
        self.tracker = None
        # grad_transport="bf16": every bucket is rounded to bf16, summed by the collective in bf16 and widened back — half the
        # bytes on the xGMI links (PyTorch's bf16_compress_hook arithmetic).  The reference all-reduces fp32 (TRAIN:87), so fp32
        # stays the default; bf16 is the documented throughput option.
        if grad_transport not in ("fp32", "bf16"):
            raise ValueError("grad_transport must be 'fp32' or 'bf16'")
        self.transport = grad_transport
        self.stage16 = None
        if self.active and grad_transport == "bf16":
            self.stage16 = torch.empty(self.arena.n, device=self.arena.grad.device, dtype=torch.bfloat16)
        if self.active and self.arena.grad.is_cuda and _COMM_MODE == "native":
            try:
                self.native = NativeComm(self.pg)
            except Exception as e:                              # (ranks sharing one card, no librccl): torch.distributed carries the data
                import sys
                print(f"[vacnic_amd.ddp] native RCCL path unavailable ({e}); using torch.distributed collectives", file=sys.stderr)
                if dist.get_backend(self.pg) == "gloo" and not _share_a_gpu(self.pg):
                    # the control-plane group must not carry 1.7 GB of gradients per step through host memory: RCCL via torch then
                    # (the failure above is symmetric — no librccl symbol, no communicator — so every rank arrives here)
                    self.pg = dist.new_group(backend="nccl")
        if self.active:
            # (i) ctor broadcast of all params from rank 0 (TRAIN:87); buffers on this path are constants
            if self.native is not None:
                from . import kernels as K
                K.comm_broadcast(self.native.h, self.arena.flat32, 0)
            else:
                dist.broadcast(self.arena.flat32, src=0, group=self.pg)
            self.arena.refresh_shadow()
            self.tracker = GradTracker(self.arena, bucket_bytes)
            self.bucket_end = dict(self.tracker.buckets)
            if overlap:
                self.tracker.on_ready = self._launch_bucket
                TRACKER = self.tracker
            if self.arena.grad.is_cuda and self.native is None:
                from . import streams as _streams
                self.comm_on_wgrad = _COMM_MODE in ("wgrad", "native") and _streams.wgrad_stream() is not None
                if not self.comm_on_wgrad:
                    self.comm_stream = torch.cuda.Stream()
            if self.native is not None:
                self.slot_of = {s: _EV_BASE + i for i, (s, _) in enumerate(self.tracker.buckets)}
                if len(self.slot_of) > 256:
                    raise ValueError("at most 256 gradient buckets (named-event slots of the library): raise bucket_bytes")

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def _launch_bucket(self, start):
        if start in self.launched:
            return
        self.launched.add(start)
        self.order.append(start)
        if self.native is not None:
            self._issue_native(start)                         # C-ABI calls only: recorded into a launch plan where they stand
        else:
            _host(lambda: self._issue_bucket(start))

    def _issue_native(self, start):
        """one bucket on the native path: the weight-gradient stream (where most of its writers ran) waits for the other writers,
        carries the (cast +) all-reduce and records the bucket's named event."""
        from . import kernels as K
        from . import streams
        end = self.bucket_end[start]
        g = self.arena.grad[start:end]
        s16 = self.stage16[start:end] if self.stage16 is not None else None
        cur = K._stream()                                     # launch stream of whatever completed the bucket (compute / branch)
        with K.launch_on(None):                               # (torch's current stream = the compute stream)
            main = K._stream()
        side = streams.wgrad_raw() if streams.wgrad_stream() is not None else main
        for src in {cur, main, streams.raw("branch") if streams.branch_stream() is not None else None}:
            if src is not None and src != side:
                K.fence(src, side)
        with K.launch_on(side, fence=False):
            if s16 is not None:
                K.cast_f32_bf16(g, s16)
            if not _SKIP_COLLECTIVE:
                K.allreduce_bucket(self.native.h, g if s16 is None else s16)
            K.event_record(self.slot_of[start])
        self.ready[start] = ("ev", self.slot_of[start], s16, g)

    def _issue_bucket(self, start):
        """host side of one bucket: the comm stream waits for the bucket's writers, then carries the (cast +) all-reduce."""
        end = self.bucket_end[start]
        g = self.arena.grad[start:end]
        s16 = self.stage16[start:end] if self.stage16 is not None else None
        if self.comm_on_wgrad:
            from . import kernels as K
            from . import streams
            side = streams.wgrad_stream()
            # the weight-gradient stream already follows most of the bucket's writers (weight-gradient GEMMs, LayerNorm parameter
            # folds); the others sit on the compute stream (embedding gradients, short LayerNorms) and on the branch stream
            side.wait_stream(torch.cuda.current_stream())
            for s in (streams.branch_stream(), torch.cuda.default_stream()):
                if s is not None:
                    side.wait_stream(s)
            w = None
            with torch.cuda.stream(side), K.launch_on(side.cuda_stream, fence=False):
                if s16 is not None:
                    K.cast_f32_bf16(g, s16)
                if not _SKIP_COLLECTIVE:
                    # the library's own stream waits for `side` as of now and carries the collective; `side` itself goes on with
                    # the next weight gradients (no wait here)
                    w = dist.all_reduce(g if s16 is None else s16, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self.ready[start] = ("work", w, s16, g)
        elif self.comm_stream is not None:
            # the bucket's writers may sit on the compute stream (LN / embedding grads) and on the weight-gradient
            # side stream (wgrad GEMMs): the collective waits for both
            from . import kernels as K
            from . import streams
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for s in (streams.wgrad_stream(), streams.branch_stream(), torch.cuda.default_stream()):
                if s is not None:
                    self.comm_stream.wait_stream(s)
            # (a bucket may complete inside a kernels.launch_on section — a branch-stream backward node: the casts below belong
            # on the comm stream whatever stream override is in force)
            with torch.cuda.stream(self.comm_stream), K.launch_on(self.comm_stream.cuda_stream, fence=False):
                if _SKIP_COLLECTIVE:                          # A/B aid: the reducer's stream structure without the library call
                    pass
                elif s16 is None:
                    w = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    w.wait()                                  # stream dependency only (NCCL work): the comm stream follows the collective
                else:
                    K.cast_f32_bf16(g, s16)
                    w = dist.all_reduce(s16, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    w.wait()
                    K.cast_bf16_f32(s16, g)
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)                   # this bucket's gradient slice holds the SUM once the event has passed
                self.ready[start] = ev
        elif s16 is None:
            self.works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            self.ready[start] = len(self.works) - 1
        else:                                                 # host arenas (gloo tests)
            s16.copy_(g)
            self.works.append((dist.all_reduce(s16, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), s16, g))
            self.ready[start] = len(self.works) - 1

    def _wait_bucket(self, start):
        """the compute stream (or, for host arenas, the host) waits for bucket `start`'s all-reduce."""
        h = self.ready[start]
        if isinstance(h, tuple) and h[0] == "ev":
            from . import kernels as K
            _, slot, s16, g = h
            K.event_wait(slot)                                # the launch (compute) stream waits for THIS bucket's all-reduce
            if s16 is not None:
                K.cast_bf16_f32(s16, g)
        elif isinstance(h, tuple):
            _, w, s16, g = h
            if w is not None:
                w.wait()                                      # the current (compute) stream waits for the collective
            if s16 is not None:
                from . import kernels as K
                K.cast_bf16_f32(s16, g)                       # widen the summed bf16 bucket back, right in front of its AdamW range
        elif isinstance(h, int):
            w = self.works[h]
            if isinstance(w, tuple):
                w[0].wait()
                w[2].copy_(w[1])
            else:
                w.wait()
        else:
            torch.cuda.current_stream().wait_event(h)

    def _finish(self):
        self.works.clear()
        self.launched.clear()
        self.order.clear()
        self.ready.clear()
        self.tracker.reset()

    def reduce_gradients(self):
        """Finish the step's gradient all-reduce: launch whatever backward did not already launch, then make the
        compute stream wait for every bucket.  Gradients hold the SUM over ranks afterwards (AdamW divides)."""
        if not self.active:
            return
        host = (lambda fn: fn()) if self.native is not None else _host
        for start, _ in self.tracker.buckets:
            self._launch_bucket(start)
        for start in list(self.order):
            host(lambda s=start: self._wait_bucket(s))
        host(self._finish)

    def reduce_and_step(self, optimizer, clip_norm=None):
        """reduce_gradients() + optimizer.step(), pipelined: AdamW runs bucket by bucket in the order the all-reduces were
        launched, each slice as soon as ITS collective has finished.  The buckets backward completes last (prompt MLP, the
        embedding tables: ~1 GB of fp32 gradients that cannot overlap with backward) are still on the links while the
        optimizer's ~5.5 ms of HBM traffic for the other ~2.5 GB runs, instead of after them.  With gradient clipping the global
        norm needs every bucket first: plain reduce, then one step."""
        if not self.active or clip_norm is not None:
            self.reduce_gradients()
            optimizer.step(clip_norm=clip_norm)
            return
        for start, _ in self.tracker.buckets:
            self._launch_bucket(start)
        host = (lambda fn: fn()) if self.native is not None else _host
        optimizer.begin_step()
        for start in list(self.order):
            host(lambda s=start: self._wait_bucket(s))
            optimizer.step_range(start, self.bucket_end[start])
        host(self._finish)
