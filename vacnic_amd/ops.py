"""Autograd operators of the VACNIC path: torch.autograd.Function shells whose forward AND backward
are hand-written HIP kernels called through the C-ABI (vacnic_amd.kernels).  torch.autograd only
records the graph; no arithmetic runs in ATen.

Conventions
  * activations are bf16, contiguous; statistics / losses fp32.
  * trainable weights live in a flat arena (vacnic_amd.arena): fp32 master + bf16 shadow + fp32
    gradient.  Weight gradients are ACCUMULATED IN PLACE into the arena by the wgrad kernels
    (split-K f32 atomics), so backward() returns None for parameters; the fp32 Parameter is still
    passed to apply() as an "anchor" so autograd tracks the op even when the data input is not
    differentiable.
  * a tensor consumed by two ops goes through fork() so the gradient fan-in is our add kernel.
"""
import torch
from torch.autograd import Function

from . import ddp
from . import kernels as K
from . import streams

BF16 = torch.bfloat16
_NO_XSUM = __import__("os").environ.get("VACNIC_NO_XSUM") == "1"
_LN_FOLD_ON_SIDE = __import__("os").environ.get("VACNIC_LN_FOLD_MAIN") != "1"      # A/B: 1 = the fold stays in the backward chain
_FUSE_ACT_DROPOUT = __import__("os").environ.get("VACNIC_FUSE_ACT_DROPOUT", "1") != "0"      # A/B: 0 = separate in-place dropout passes


class Rng:
    """Dropout seeds: one fresh 64-bit seed per dropout site per step (Philox key); the kernels
    regenerate masks from (seed, element index) in backward, nothing is stored."""
    base = 0x5EED
    counter = 0

    @classmethod
    def manual_seed(cls, s):
        cls.base = int(s) & 0xFFFFFFFF
        cls.counter = 0

    @classmethod
    def next(cls):
        cls.counter += 1
        return (cls.base << 32) | (cls.counter & 0xFFFFFFFF)

    dev = None      # int64 [1] on the GPU: per-step counter mixed into every site seed (incremented by lr_step)

    @classmethod
    def device_counter(cls):
        if cls.dev is None:
            cls.dev = torch.zeros(1, dtype=torch.int64, device="cuda")
        return cls.dev


class LinearSpec:
    """What a GEMM needs to know about an nn.Linear: bf16 shadow weight [N,K], fp32 bias, grad views."""
    __slots__ = ("w16", "bias", "wgrad", "bgrad", "N", "K", "ldw")

    def __init__(self, w16, bias=None, wgrad=None, bgrad=None):
        self.w16, self.bias, self.wgrad, self.bgrad = w16, bias, wgrad, bgrad
        self.N, self.K = w16.shape
        self.ldw = w16.stride(0)


def _c(t):
    """t, contiguous.  A strided bf16 view (unit inner stride, <= 3-D) is packed by OUR copy kernel on the op's launch stream:
    Tensor.contiguous() would be an ATen kernel on torch's current stream — not the side stream a backward node launches on, and
    invisible to a recorded launch plan (a replay would then read stale data)."""
    if t.is_contiguous():
        return t
    if t.is_cuda and t.dtype == BF16 and t.stride(-1) == 1 and 2 <= t.dim() <= 3:
        t3 = t if t.dim() == 3 else t.unsqueeze(0)
        out = torch.empty(t3.shape, device=t.device, dtype=BF16)
        K.copy3d(t3, out, t3.shape[0], t3.shape[1], t3.shape[2])
        return out.view(t.shape)
    if K._OVERRIDE is not None or K.recording():
        raise RuntimeError(f"vacnic_amd.ops: cannot pack a {t.dtype} view with strides {t.stride()} without an ATen kernel "
                           "(side-stream launch or launch-plan recording in progress)")
    return t.contiguous()


def _bw(fn):
    """decorator of Function.backward: launch on the HIP stream the op's forward was launched on (ctx.raw, set by _tag).  With
    explicit scheduling torch sees one stream, so the autograd engine no longer puts a node's backward on its forward's stream
    (nor synchronises around it: the fences are ours, StreamHopFn)."""
    def backward(ctx, *grads):
        raw = getattr(ctx, "raw", None)
        if raw is None or raw == K._OVERRIDE:
            return fn(ctx, *grads)
        with K.launch_on(raw):
            return fn(ctx, *grads)
    return staticmethod(backward)


def _tag(ctx):
    ctx.raw = K._OVERRIDE
    # an output nobody differentiates (a decoder state only the caller keeps, an unused skip branch) hands None to backward
    # instead of a zero tensor autograd would fill with an ATen kernel — 13 fills per step, and kernels the launch-plan recorder
    # cannot see.  Every backward here treats None as "no contribution".
    ctx.set_materialize_grads(False)


class StreamHopFn(Function):
    """identity on tensors that cross from HIP stream `src` (where they were produced) to stream `dst` (where they are consumed):
    forward = fence(src -> dst); backward = fence(dst -> src), issued by the autograd engine exactly when all the gradients of the
    crossing tensors have been enqueued on dst.  The explicit counterpart of what the engine does implicitly for ops run under
    torch.cuda.stream()."""

    @staticmethod
    def forward(ctx, src, dst, *xs):
        ctx.hop = (src, dst)
        ctx.set_materialize_grads(False)           # an output nobody differentiates keeps a None gradient (no zero-fill kernel)
        K.fence(src, dst)
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        src, dst = ctx.hop
        K.fence(dst, src)
        return (None, None) + gs


def stream_hop(src, dst, *xs):
    """xs, usable on stream dst (see StreamHopFn).  Tensors that need no gradient are fenced without a graph node."""
    if src == dst:
        return xs
    if torch.is_grad_enabled() and any(x.requires_grad for x in xs):
        return StreamHopFn.apply(src, dst, *xs)
    K.fence(src, dst)
    return xs


# ---- grouped weight gradients -------------------------------------------------------------------------------------------
# A weight gradient with a long reduction (M rows) and a small output (d x d .. 4d x d) cannot fill 256 CUs without split-K,
# and split-K sums through fp32 atomics.  Weight gradients are needed only by AdamW / the DDP reducer, so they can be DEFERRED:
# jobs queue up during backward and leave in groups of WGRAD_GROUP_UNITS output blocks of 1024 x 1024, each block with its full
# reduction on one XCD (vacnic_wgrad_group): no atomics on dW, bitwise reproducible sums, a fraction of the launches.  The queue
# empties at the end of backward (autograd engine callback).
# WHICH jobs are grouped is decided by the whole step, not by the kernel alone (same-box A/B, profiles/r3_step_ab_wgrad.txt):
#   grouped everything 67.4 ms/step | split-K everything (round 2) 66.6 | grouped for M <= 4096, split-K above: 65.8
# Alone, the grouped kernel wins on every shape (d x d at M = 16384: 68 -> 41 us), but the weight gradients share the GPU with the
# compute stream's dgrad chain, and the encoder-sized groups (16 blocks x 16384 rows: 0.6 ms of every CU) delay that chain by
# more than they save; the decoder-sized jobs (M = B*T = 2048: 1.8-3.9x faster grouped, 72 launches -> 12) are the clear win.
# VACNIC_WGRAD_GROUP=0: one split-K GEMM per Linear.  VACNIC_WGRAD_GROUP_MAX_M=<rows>: group up to that reduction length
# (a very large value = every weight gradient grouped: the deterministic mode).
WGRAD_GROUP = __import__("os").environ.get("VACNIC_WGRAD_GROUP", "1") != "0"
WGRAD_GROUP_UNITS = int(__import__("os").environ.get("VACNIC_WGRAD_GROUP_UNITS", "16"))     # queued output blocks that trigger a launch
WGRAD_GROUP_MIN_M = 1024
WGRAD_GROUP_MAX_M = int(__import__("os").environ.get("VACNIC_WGRAD_GROUP_MAX_M", "4096"))


class _WgradQueue:
    def __init__(self):
        self.by_m = {}            # reduction length -> [(dy2d, x2d, spec)]
        self.units = {}
        self.dst = set()          # gradient views with a queued job (a second writer of the same view must not share a launch)
        self.armed = False

    def add(self, dy2d, x2d, spec, M):
        if spec.wgrad.data_ptr() in self.dst:          # tied weights (--init_attn_weight): serialise the writers
            self.flush()
        if not self.armed:
            # the engine runs this once, after the last node of the CURRENT backward pass
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
                self.armed = True
            except RuntimeError:                       # not inside a backward pass (a direct call): nothing will flush later
                self.by_m.setdefault(M, []).append((dy2d, x2d, spec))
                self._launch(M)
                return
        self.dst.add(spec.wgrad.data_ptr())
        self.by_m.setdefault(M, []).append((dy2d, x2d, spec))
        u = self.units[M] = self.units.get(M, 0) + ((spec.N + 1023) // 1024) * ((spec.K + 1023) // 1024)
        if u >= WGRAD_GROUP_UNITS:
            self._launch(M)

    def _launch(self, M):
        jobs = self.by_m.pop(M, [])
        self.units.pop(M, None)
        if not jobs:
            return
        for _, _, sp in jobs:
            self.dst.discard(sp.wgrad.data_ptr())
        packed = [(dy, x, sp.wgrad, sp.bgrad) for dy, x, sp in jobs]
        side = streams.wgrad_stream()
        if side is None:
            K.wgrad_group(packed)
            for _, _, sp in jobs:
                ddp.done(sp.wgrad, sp.bgrad)
            return
        # (the side stream already waits for every queued job's producer: add() is called behind a producer -> side fence)
        if ddp.TRACKER is None or streams.explicit():
            with K.launch_on(streams.wgrad_raw(), fence=False):      # (every job fenced its producer -> wgrad stream when it was queued)
                K.wgrad_group(packed)
            for _, _, sp in jobs:
                ddp.done(sp.wgrad, sp.bgrad)
        else:
            with torch.cuda.stream(side):
                K.wgrad_group(packed)
                for _, _, sp in jobs:
                    ddp.done(sp.wgrad, sp.bgrad)
        for dy, x, _ in jobs:
            streams.keep(dy, x)

    def flush(self):
        self.armed = False
        for M in list(self.by_m):
            self._launch(M)

    def reset(self):
        """forget everything queued: a backward pass that RAISED never runs the engine's final callbacks, so `armed` would stay
        set and the failed pass's jobs (stale dy / x tensors) would ride into the next step's launches.  train_step calls this
        before every forward and flush() right after backward()."""
        self.by_m.clear()
        self.units.clear()
        self.dst.clear()
        self.armed = False


_WGQ = _WgradQueue()


def flush_wgrads():
    """launch every deferred weight gradient now (the autograd engine does this at the end of backward)."""
    _WGQ.flush()


def begin_step():
    """start of a training step: drop whatever a failed previous step left in the deferred weight-gradient queue, and release the
    side-stream keep-list behind a join if that step never reached its own (an aborted step, a grad-enabled forward without a
    backward: the list would otherwise pin every branch / aux / comm-stream activation)."""
    _WGQ.reset()
    if streams.pending_keep():
        streams.join_all()


def _groupable(dy2d, x2d, spec, M):
    return (WGRAD_GROUP and spec.wgrad is not None and not _NO_XSUM and WGRAD_GROUP_MIN_M <= M <= WGRAD_GROUP_MAX_M and spec.N >= 512 and spec.K >= 512
            and dy2d.stride(0) % 8 == 0 and x2d.stride(0) % 8 == 0 and spec.wgrad.stride(0) % 4 == 0
            and dy2d.shape[1] == spec.N and x2d.shape[1] == spec.K)


def _wgrad(dy2d, x2d, spec, M):
    """spec.wgrad[N,K] += dy^T x ; spec.bgrad[N] += colsum(dy).  Off the critical path: issued on the weight-gradient
    side stream when streams are enabled (each weight has exactly one writer op, so there is no cross-stream race)."""
    side = streams.wgrad_stream()
    group = _groupable(dy2d, x2d, spec, M)
    if side is None:
        if group:
            _WGQ.add(dy2d, x2d, spec, M)
            return
        _wgrad_impl(dy2d, x2d, spec, M)
        ddp.done(spec.wgrad, spec.bgrad)
        return
    if streams.explicit():
        K.fence(K._stream(), streams.wgrad_raw())          # dy / x were produced on the stream this backward node launches on
    else:
        side.wait_stream(torch.cuda.current_stream())      # dy / x were produced on the compute stream
    if group:
        _WGQ.add(dy2d, x2d, spec, M)
        return
    if ddp.TRACKER is None or streams.explicit():
        with K.launch_on(streams.wgrad_raw(), fence=False):
            _wgrad_impl(dy2d, x2d, spec, M)
        ddp.done(spec.wgrad, spec.bgrad)
    else:
        with torch.cuda.stream(side):                      # the bucket launcher reads torch's current stream
            _wgrad_impl(dy2d, x2d, spec, M)
            ddp.done(spec.wgrad, spec.bgrad)
    streams.keep(dy2d, x2d)                                # alive until the compute stream joins the side stream


# split-K weight gradients through the GEMM's ordered fix-up instead of fp32 atomics (bitwise reproducible dW):
#   0 = atomics (rounds 1-3); 1 = fix-up with the same 128 x 128 tiles and split factors; 2 = fix-up, 256 x 256 tiles x 4 slices for
#   outputs larger than 1024 x 1024 (the isolated winners of profiles/r4_gemm_fixup_vs_shipped.txt)
# Default 2: same-box A/B 66.79 vs 67.16 ms/step (4 of 4 interleaved runs, profiles/r4_step_ab_fixup.txt) — and with it no weight
# gradient is summed through atomics any more (the decoder-sized ones go through the grouped kernel): dW is bitwise reproducible.
WGRAD_FIXUP = int(__import__("os").environ.get("VACNIC_WGRAD_FIXUP", "2"))


def _wgrad_impl(dy2d, x2d, spec, M):
    if spec.wgrad is not None and not _NO_XSUM:
        N, Kd = spec.N, spec.K
        tiles = ((N + 127) // 128) * ((Kd + 127) // 128)
        split = K.wgrad_split(M, tiles)
        if WGRAD_FIXUP and split > 1 and M >= 4096:
            hint = 128
            if WGRAD_FIXUP == 2 and N >= 512 and Kd >= 512:
                hint, split = (128, 8) if N * Kd <= (1 << 20) else (256, 4)
            K.gemm(dy2d, x2d, N, Kd, M, out=spec.wgrad, ldx=dy2d.stride(0), ldw=x2d.stride(0), ldo=spec.wgrad.stride(0),
                   x_kstrided=True, w_kstrided=True, out_mode=2, split_k=split, xsum=spec.bgrad, fixup=True, tile_hint=hint)
            return
        # the bias gradient (column sums of dY) rides on the weight-gradient GEMM's own dY fragments (xsum): no second pass
        K.gemm(dy2d, x2d, N, Kd, M, out=spec.wgrad, ldx=dy2d.stride(0), ldw=x2d.stride(0), ldo=spec.wgrad.stride(0),
               x_kstrided=True, w_kstrided=True, out_mode=2, split_k=split, xsum=spec.bgrad)
    else:
        if spec.wgrad is not None:                  # A/B: VACNIC_NO_XSUM=1 restores the separate bias-gradient reduction
            N, Kd = spec.N, spec.K
            tiles = ((N + 127) // 128) * ((Kd + 127) // 128)
            K.gemm(dy2d, x2d, N, Kd, M, out=spec.wgrad, ldx=dy2d.stride(0), ldw=x2d.stride(0), ldo=spec.wgrad.stride(0),
                   x_kstrided=True, w_kstrided=True, out_mode=2, split_k=K.wgrad_split(M, tiles))
        if spec.bgrad is not None:
            K.bias_grad(dy2d, spec.bgrad, M, spec.N)


# ------------------------------------------------------------------------------------------- Linear
class LinearFn(Function):
    """y = x W^T + b (+ residual).  nn.Linear on the path: q/kv/out projections, visual_map, _linear_1.
    skip=True: also returns x itself as a second output (the residual branch of a post-LN block, MFULL:697,711): the
    gradient fan-in of x — dgrad of this Linear + the gradient arriving on the skip branch — is then the `residual` input of
    the dgrad GEMM's epilogue instead of a separate add kernel over [B*S, d]."""

    @staticmethod
    def forward(ctx, x, anchor, spec, residual, ge, skip):
        _tag(ctx)
        Kd = spec.K
        x2 = _c(x).view(-1, Kd)
        M = x2.shape[0]
        out = torch.empty(x.shape[:-1] + (spec.N,), device=x.device, dtype=BF16)
        K.gemm(x2, spec.w16, M, spec.N, Kd, bias=spec.bias, out=out, ldw=spec.ldw,
               residual=_c(residual) if residual is not None else None)
        ctx.spec, ctx.M = spec, M
        ctx.has_res = residual is not None
        ctx.save_for_backward(x2)
        ddp.expect(ge and any(ctx.needs_input_grad), spec.wgrad, spec.bgrad)
        if skip:
            return out, x.view_as(x)
        return out

    @_bw
    def backward(ctx, dy, dskip=None):
        (x2,) = ctx.saved_tensors
        spec, M = ctx.spec, ctx.M
        if dy is None:                        # only the skip output was differentiated: this Linear contributes nothing
            ddp.done(spec.wgrad, spec.bgrad)
            return dskip, None, None, None, None, None
        dy2 = _c(dy).view(M, spec.N)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, spec.K), device=dy.device, dtype=BF16)
            K.gemm(dy2, spec.w16, M, spec.K, spec.N, out=dx, ldw=spec.ldw, w_kstrided=True,
                   residual=_c(dskip).view(M, spec.K) if dskip is not None else None)
            dx = dx.view(dy.shape[:-1] + (spec.K,))
        _wgrad(dy2, x2, spec, M)
        return dx, None, None, (dy if ctx.has_res else None), None, None


def linear(x, anchor, spec, residual=None):
    return LinearFn.apply(x, anchor, spec, residual, torch.is_grad_enabled(), False)


def linear_skip(x, anchor, spec):
    """(linear(x), x): the skip output replaces ops.fork(x) when x's other consumer starts with this Linear."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return linear(x, anchor, spec), x
    return LinearFn.apply(x, anchor, spec, None, True, True)


# --------------------------------------------------------------------------------------------- MLP2
class Mlp2Fn(Function):
    """y = W2 dropout(act(W1 x + b1), p_act) + b2 — every two-layer block on the path: text/img/face FFN (MFULL:647-664,
    738-741), name-prefix FFN on the flat view (:682-687), ClipCap prompt MLP (MFULL:111-123), ViT MLP.
    The activation backward is fused into the dgrad GEMM epilogue (dact_src).  p_act > 0 (config.activation_dropout; 0.1 in the
    bart-base / bart-large hub configs): an in-place Philox dropout pass over the hidden activations, repeated on their gradient in backward."""

    @staticmethod
    def forward(ctx, x, anchor, s1, s2, act, ge, skip=False, p_act=0.0, seed=0):
        _tag(ctx)
        x2 = _c(x).view(-1, s1.K)
        M = x2.shape[0]
        need = ge and any(ctx.needs_input_grad)      # grad mode is always off inside Function.forward: `ge` comes from the caller
        u = torch.empty((M, s1.N), device=x.device, dtype=BF16) if need else None
        h = torch.empty((M, s1.N), device=x.device, dtype=BF16)
        # activation dropout rides in the epilogue (after the activation); widths that are not multiples of 16: a separate pass
        fuse = p_act > 0.0 and K.can_fuse_dropout(s1.N) and _FUSE_ACT_DROPOUT
        K.gemm(x2, s1.w16, M, s1.N, s1.K, bias=s1.bias, out=h, ldw=s1.ldw, act=act, preact=u,
               drop=(p_act, seed, Rng.device_counter()) if fuse else None)
        if p_act > 0.0 and not fuse:
            K.dropout_(h, p_act, seed, Rng.device_counter())
        out = torch.empty(x.shape[:-1] + (s2.N,), device=x.device, dtype=BF16)
        K.gemm(h, s2.w16, M, s2.N, s2.K, bias=s2.bias, out=out, ldw=s2.ldw)
        ctx.s1, ctx.s2, ctx.act, ctx.M = s1, s2, act, M
        ctx.p_act, ctx.seed = p_act, seed
        ctx.save_for_backward(x2, u, h)
        ddp.expect(need, s1.wgrad, s1.bgrad, s2.wgrad, s2.bgrad)
        if skip:
            return out, x.view_as(x)             # residual branch: its gradient joins in the last dgrad GEMM's epilogue
        return out

    @_bw
    def backward(ctx, dy, dskip=None):
        x2, u, h = ctx.saved_tensors
        s1, s2, act, M = ctx.s1, ctx.s2, ctx.act, ctx.M
        if dy is None:                        # only the skip output was differentiated
            ddp.done(s1.wgrad, s1.bgrad, s2.wgrad, s2.bgrad)
            return dskip, None, None, None, None, None, None, None, None
        dy2 = _c(dy).view(M, s2.N)
        if s2.N % 8:                                   # 20-wide name-prefix output: give the GEMMs 16-byte rows
            dy2 = K.pad_cols(dy2, (s2.N + 7) // 8 * 8)
        du = torch.empty((M, s1.N), device=dy.device, dtype=BF16)
        fuse = ctx.p_act > 0.0 and K.can_fuse_dropout(s1.N) and _FUSE_ACT_DROPOUT
        K.gemm(dy2, s2.w16, M, s1.N, s2.N, out=du, ldx=dy2.stride(0), ldw=s2.ldw, w_kstrided=True, act=act, dact_src=u,
               drop=(ctx.p_act, ctx.seed, Rng.device_counter()) if fuse else None)
        if ctx.p_act > 0.0 and not fuse:               # the mask commutes with the elementwise act'(u) the epilogue applied
            K.dropout_(du, ctx.p_act, ctx.seed, Rng.device_counter())
        _wgrad(dy2, h, s2, M)                          # h is the DROPPED activation: what fc2 saw in forward
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, s1.K), device=dy.device, dtype=BF16)
            K.gemm(du, s1.w16, M, s1.K, s1.N, out=dx, ldw=s1.ldw, w_kstrided=True,
                   residual=_c(dskip).view(M, s1.K) if dskip is not None else None)
            dx = dx.view(dy.shape[:-1] + (s1.K,))
        _wgrad(du, x2, s1, M)
        return dx, None, None, None, None, None, None, None, None


def mlp2(x, anchor, s1, s2, act="gelu", p_act=0.0, training=False):
    p = p_act if training else 0.0
    return Mlp2Fn.apply(x, anchor, s1, s2, act, torch.is_grad_enabled(), False, p, Rng.next() if p > 0 else 0)


def mlp2_skip(x, anchor, s1, s2, act="gelu", p_act=0.0, training=False):
    """(mlp2(x), x) — see linear_skip."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return mlp2(x, anchor, s1, s2, act, p_act, training), x
    p = p_act if training else 0.0
    return Mlp2Fn.apply(x, anchor, s1, s2, act, True, True, p, Rng.next() if p > 0 else 0)


class MlpChainFn(Function):
    """y = W_n act(... act(W_1 x + b_1) ...) + b_n — the token-mixing prompt MLP of `--prompt_mlp_type mlp` (MFULL:76-108):
    any number of Linear layers with `act` between them.  The first layer's input width (the ViT's patch-token count, 196 for
    ViT-B/16) need not be a multiple of 8: x and W_1 are repacked to 16-byte rows (zero columns contribute nothing), and the
    weight gradient is accumulated straight into the unpadded [N, K] gradient view.  Hidden widths are multiples of 8
    (VacnicConfig.validate).  Activation backward is fused into the dgrad epilogue of the following layer (dact_src)."""

    @staticmethod
    def forward(ctx, x, anchor, specs, act, ge):
        _tag(ctx)
        s0 = specs[0]
        x2 = _c(x).view(-1, s0.K)
        M = x2.shape[0]
        need = ge and any(ctx.needs_input_grad)
        Kp = (s0.K + 7) // 8 * 8
        w0 = s0.w16
        if Kp != s0.K:
            x2 = K.pad_cols(x2, Kp)
            w0 = K.pad_cols(w0, Kp)
        elif s0.ldw % 8:
            w0 = K.pad_cols(w0, Kp)
        saved, h = [x2], x2
        for i, sp in enumerate(specs):
            last = i == len(specs) - 1
            u = torch.empty((M, sp.N), device=x.device, dtype=BF16) if (need and not last) else None
            o = torch.empty((M, sp.N), device=x.device, dtype=BF16)
            w, ldw, kd = (w0, w0.stride(0), Kp) if i == 0 else (sp.w16, sp.ldw, sp.K)
            K.gemm(h, w, M, sp.N, kd, bias=sp.bias, out=o, ldx=h.stride(0), ldw=ldw, act=None if last else act, preact=u)
            if not last:
                saved += [u, o]
            h = o
        ctx.specs, ctx.act, ctx.M, ctx.xshape = specs, act, M, x.shape
        ctx.save_for_backward(*saved)
        for sp in specs:
            ddp.expect(need, sp.wgrad, sp.bgrad)
        return h

    @_bw
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        specs, act, M = ctx.specs, ctx.act, ctx.M
        d = _c(dy).view(M, specs[-1].N)
        for i in range(len(specs) - 1, -1, -1):
            sp = specs[i]
            inp = saved[0] if i == 0 else saved[2 * i]              # this layer's input: x, or the previous activation output
            if i > 0:
                u_prev = saved[2 * i - 1]
                dprev = torch.empty((M, sp.K), device=dy.device, dtype=BF16)
                K.gemm(d, sp.w16, M, sp.K, sp.N, out=dprev, ldx=d.stride(0), ldw=sp.ldw, w_kstrided=True, act=act, dact_src=u_prev)
            _wgrad(d, inp, sp, M)                                   # layer 0: inp is the 16-byte-row repack; only the first K columns are written
            if i > 0:
                d = dprev
        dx = None
        if ctx.needs_input_grad[0]:
            s0 = specs[0]
            if s0.K % 8:
                raise NotImplementedError("input gradient through a ragged-K first layer (the image features are frozen: TRAIN:274-276)")
            dx = torch.empty((M, s0.K), device=dy.device, dtype=BF16)
            K.gemm(d, s0.w16, M, s0.K, s0.N, out=dx, ldw=s0.ldw, w_kstrided=True)
            dx = dx.view(ctx.xshape)
        return dx, None, None, None, None


def mlp_chain(x, anchor, specs, act="tanh"):
    return MlpChainFn.apply(x, anchor, tuple(specs), act, torch.is_grad_enabled())


# ---------------------------------------------------------------------------------------- attention
class SelfAttnFn(Function):
    """kvq: [B,T,3d] fused projection output, column blocks [K | V | Q] (arena order k,v,q).  p_drop: attention-probability dropout
    (MFULL:546), Philox mask regenerated in backward from the same seed."""

    @staticmethod
    def forward(ctx, kvq, key_mask, causal, H, ge, p_drop=0.0, seed=0):
        _tag(ctx)
        B, T, d3 = kvq.shape
        d = d3 // 3
        sd = Rng.device_counter() if p_drop > 0 else None
        out, lse = K.attn_fwd(kvq[..., 2 * d:], kvq[..., :d], kvq[..., d:2 * d], B, H, T, T, key_mask=key_mask,
                              causal=causal, scale=0.125, need_lse=ge and any(ctx.needs_input_grad), p_drop=p_drop, seed=seed, seed_dev=sd)
        ctx.cfg = (B, H, T, d, causal)
        ctx.key_mask = key_mask
        ctx.drop = (p_drop, seed)
        ctx.save_for_backward(kvq, out, lse)
        return out

    @_bw
    def backward(ctx, dout):
        kvq, out, lse = ctx.saved_tensors
        B, H, T, d, causal = ctx.cfg
        p_drop, seed = ctx.drop
        dkvq = torch.empty_like(kvq)
        K.attn_bwd(kvq[..., 2 * d:], kvq[..., :d], kvq[..., d:2 * d], out, _c(dout), lse, dkvq[..., 2 * d:], dkvq[..., :d],
                   dkvq[..., d:2 * d], B, H, T, T, key_mask=ctx.key_mask, causal=causal, scale=0.125, p_drop=p_drop, seed=seed,
                   seed_dev=Rng.device_counter() if p_drop > 0 else None)
        return dkvq, None, None, None, None, None, None


class CrossAttnFn(Function):
    """q: [B,Tq,d]; kv: [B,Tk,2d] = [K | V] (any row stride: e.g. one layer's slice of the batched decoder K|V buffer).
    bank / slot: when kv is output `slot` of SplitKvFn, dK|dV are written straight into the bank's [B,Tk,L*2d] gradient buffer."""

    @staticmethod
    def forward(ctx, q, kv, key_mask, H, ge, bank=None, slot=0, p_drop=0.0, seed=0):
        _tag(ctx)
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        sd = Rng.device_counter() if p_drop > 0 else None
        out, lse = K.attn_fwd(q, kv[..., :d], kv[..., d:], B, H, Tq, Tk, key_mask=key_mask, causal=False, scale=0.125,
                              need_lse=ge and any(ctx.needs_input_grad), p_drop=p_drop, seed=seed, seed_dev=sd)
        ctx.cfg = (B, H, Tq, Tk, d)
        ctx.key_mask = key_mask
        ctx.bank, ctx.slot = bank, slot
        ctx.drop = (p_drop, seed)
        ctx.save_for_backward(q, kv, out, lse)
        return out

    @_bw
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        B, H, Tq, Tk, d = ctx.cfg
        p_drop, seed = ctx.drop
        dq = torch.empty_like(q)
        if ctx.bank is not None:
            dkv = ctx.bank.grad_slot(ctx.slot)
        else:
            dkv = torch.empty((B, Tk, 2 * d), device=kv.device, dtype=BF16)
        K.attn_bwd(q, kv[..., :d], kv[..., d:], out, _c(dout), lse, dq, dkv[..., :d], dkv[..., d:], B, H, Tq, Tk,
                   key_mask=ctx.key_mask, causal=False, scale=0.125, p_drop=p_drop, seed=seed,
                   seed_dev=Rng.device_counter() if p_drop > 0 else None)
        return dq, dkv, None, None, None, None, None, None, None


class KvBank:
    """The batched K|V projection of ONE source for L cross-attentions ([B, Tk, L*2d], layer l at columns [l*2d, (l+1)*2d)) and,
    in backward, the gradient buffer of the same layout that the L attention backwards fill slot by slot."""

    def __init__(self, kv_all, L):
        self.kv_all, self.L = kv_all, L
        self.w = kv_all.shape[-1] // L
        self.dkv_all = None

    def slot(self, t, l):
        return t[..., l * self.w:(l + 1) * self.w]

    def grad_slot(self, l):
        if self.dkv_all is None:
            self.dkv_all = torch.empty_like(self.kv_all)
        return self.slot(self.dkv_all, l)


class SplitKvFn(Function):
    """kv_all [B,Tk,L*2d] -> L strided views.  Backward: each incoming gradient is (normally) already the matching slot of the
    bank's gradient buffer, written in place by CrossAttnFn.backward, so the fan-in is free; anything else is copied in."""

    @staticmethod
    def forward(ctx, kv_all, bank):
        _tag(ctx)
        ctx.bank = bank
        return tuple(bank.slot(kv_all, l) for l in range(bank.L))

    @_bw
    def backward(ctx, *grads):
        bank = ctx.bank
        for l, g in enumerate(grads):
            dst = bank.grad_slot(l)
            if g is None:
                K.zero_strided(dst)
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                K.copy3d(_c(g), dst, g.shape[0], g.shape[1], g.shape[2])
        out = bank.dkv_all
        bank.dkv_all = None
        return out, None


def split_kv(kv_all, L):
    """(views, bank) for L cross-attentions over one batched K|V projection."""
    bank = KvBank(kv_all, L)
    if not (torch.is_grad_enabled() and kv_all.requires_grad):
        return [bank.slot(kv_all, l) for l in range(L)], None
    return list(SplitKvFn.apply(kv_all, bank)), bank


def self_attention(kvq, key_mask, causal, H, p_drop=0.0, training=False):
    p = p_drop if training else 0.0
    return SelfAttnFn.apply(kvq, key_mask, causal, H, torch.is_grad_enabled(), p, Rng.next() if p > 0 else 0)


def cross_attention(q, kv, key_mask, H, bank=None, slot=0, p_drop=0.0, training=False):
    p = p_drop if training else 0.0
    return CrossAttnFn.apply(q, kv, key_mask, H, torch.is_grad_enabled(), bank, slot, p, Rng.next() if p > 0 else 0)


# -------------------------------------------------------------------------------------------- LN family
class AddLnFn(Function):
    """LayerNorm(residual + dropout(x)); residual may be None (plain LN)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, p_drop, seed, ge):
        _tag(ctx)
        x = _c(x)
        res = _c(residual) if residual is not None else None
        need = ge and any(ctx.needs_input_grad)
        sd = Rng.device_counter() if p_drop > 0 else None
        out, mean, rstd = K.add_ln_fwd(x, res, gamma, beta, p_drop=p_drop, seed=seed, need_stats=need, seed_dev=sd)
        ctx.p, ctx.seed = p_drop, seed
        ctx.gb = (gamma, beta)
        ctx.has_res = res is not None
        ctx.save_for_backward(x, res, mean, rstd)
        ddp.expect(need, gamma.grad, beta.grad)
        return out

    @_bw
    def backward(ctx, dout):
        x, res, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.gb
        # the fold of the LayerNorm parameter gradients (needed by AdamW / the reducer only) rides on the weight-gradient stream
        side = streams.wgrad_raw() if (streams.explicit() and streams.wgrad_stream() is not None and _LN_FOLD_ON_SIDE) else None
        dx, dres = K.add_ln_bwd(_c(dout), x, res, gamma, mean, rstd, gamma.grad, beta.grad, p_drop=ctx.p, seed=ctx.seed,
                                seed_dev=Rng.device_counter() if ctx.p > 0 else None,
                                need_dres=ctx.has_res and ctx.needs_input_grad[1], fold_on=side)
        ddp.done(gamma.grad, beta.grad)
        return (dx if ctx.needs_input_grad[0] else None), (dres if ctx.has_res and ctx.needs_input_grad[1] else None), None, None, None, None, None


def add_ln(x, residual, gamma, beta, p_drop=0.0, training=False):
    p = p_drop if training else 0.0
    return AddLnFn.apply(x, residual, gamma, beta, p, Rng.next() if p > 0 else 0, torch.is_grad_enabled())


class EmbedLnFn(Function):
    """dropout(LayerNorm(embed[ids]*scale + pos[t+2])) (MFULL:1243-1249)."""

    @staticmethod
    def forward(ctx, ids, tok, pos, gamma, beta, scale, p_drop, seed, padding_idx, ge):
        _tag(ctx)
        out, mean, rstd = K.embed_ln_fwd(ids, tok.w16, pos.w16, gamma, beta, embed_scale=scale, p_drop=p_drop, seed=seed,
                                         seed_dev=Rng.device_counter() if p_drop > 0 else None)
        ctx.args = (tok, pos, gamma, beta, scale, p_drop, seed, padding_idx)
        ctx.save_for_backward(ids, mean, rstd)
        ddp.expect(ge and any(ctx.needs_input_grad), tok.grad, pos.grad, gamma.grad, beta.grad)
        return out

    @_bw
    def backward(ctx, dout):
        ids, mean, rstd = ctx.saved_tensors
        tok, pos, gamma, beta, scale, p, seed, pad = ctx.args
        if streams.explicit() and streams.wgrad_stream() is not None:
            # the tied matrix has another writer on the weight-gradient stream (the LM head's dE GEMMs: plain read-modify-write of
            # the same rows this kernel adds to with atomics): order behind it
            K.fence(streams.wgrad_raw(), K._stream())
        K.embed_ln_bwd(ids, tok.w16, pos.w16, _c(dout), gamma, mean, rstd, tok.grad, pos.grad, gamma.grad, beta.grad,
                       embed_scale=scale, padding_idx=pad, p_drop=p, seed=seed, seed_dev=Rng.device_counter() if p > 0 else None)
        ddp.done(tok.grad, pos.grad, gamma.grad, beta.grad)
        return (None,) * 10


def embed_ln(ids, tok, pos, gamma, beta, scale=1.0, p_drop=0.0, training=False, padding_idx=1):
    p = p_drop if training else 0.0
    return EmbedLnFn.apply(ids, tok, pos, gamma, beta, scale, p, Rng.next() if p > 0 else 0, padding_idx, torch.is_grad_enabled())


# -------------------------------------------------------------------------------------------- plumbing
class ForkFn(Function):
    """identity with two consumers; backward = our bf16 add of the two incoming gradients."""

    @staticmethod
    def forward(ctx, x):
        _tag(ctx)
        return x.view_as(x), x.view_as(x)

    @_bw
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return K.add(_c(g1), _c(g2))


def fork(x):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    return ForkFn.apply(x)


class CatTokensFn(Function):
    """torch.cat(parts, dim=1) (MFULL:668,691) as strided copies; backward slices by strided copies."""

    @staticmethod
    def forward(ctx, *parts):
        _tag(ctx)
        ctx.lens = [p.shape[1] for p in parts]
        return K.cat_tokens([_c(p) for p in parts])

    @_bw
    def backward(ctx, g):
        g = _c(g)
        B, _, D = g.shape
        outs, o = [], 0
        for i, n in enumerate(ctx.lens):
            if ctx.needs_input_grad[i]:
                t = torch.empty((B, n, D), device=g.device, dtype=BF16)
                K.copy3d(g[:, o:o + n], t, B, n, D)
                outs.append(t)
            else:
                outs.append(None)
            o += n
        return tuple(outs)


def cat_tokens(*parts):
    return CatTokensFn.apply(*parts)


class CastInFn(Function):
    """fp32 input -> bf16 activation (no gradient: inputs are data)."""

    @staticmethod
    def forward(ctx, x):
        _tag(ctx)
        return K.cast_f32_bf16(_c(x))

    @_bw
    def backward(ctx, g):
        return None


def to_bf16(x):
    return x if x.dtype == BF16 else CastInFn.apply(x)


# ---------------------------------------------------------------------------------------------- losses
LMHEAD_CHUNK = 16384        # vocabulary columns of dlogits alive at a time in the backward ([R, 16384] bf16 = 64 MiB at R = 2048)
# dh += dlogits_c . E_c reduces over 16384 vocabulary columns into a [R, d] output: split-K.  Default: the K slices meet through the
# GEMM's ordered fix-up (workspace + arrival tickets, include/vacnic_hip.h) — bitwise reproducible at the speed of the fp32
# atomics it replaces (R = 2048: 89.7 vs 88.0 us per chunk, profiles/r4_gemm_fixup_vs_shipped.txt).  With atomics the summation
# order varied whenever a second process shared the GPU, one bf16 element of dh rounded the other way about every second pass,
# and backward amplified that to 1e-5 .. 2e-3 on every gradient (tools/grad_determinism.py).  VACNIC_LMHEAD_ATOMICS=1: round 3.
LMHEAD_FIXUP = __import__("os").environ.get("VACNIC_LMHEAD_ATOMICS") != "1"


class LmHeadCeFn(Function):
    """lm_head (tied to shared, MFULL:1885,1997) + CrossEntropyLoss(ignore_index=pad) (TRAIN:287) WITHOUT the [R, V] logits
    (SURVEY §8a row a8).  Forward: one GEMM whose epilogue keeps, per 256-column tile and row, the online-softmax pair and
    the target logit (3 MB instead of 412 MB of fp32 logits) + a combine kernel.  Backward: the logits are recomputed one
    vocabulary chunk at a time directly as bf16 dlogits, each chunk feeding the dh (split-K, fp32 accumulate) and dE
    (into the tied matrix's gradient rows) GEMMs before the buffer is re-used."""

    @staticmethod
    def forward(ctx, h, anchor, emb16_pad, egrad, targets, V, ignore_index, ge):
        _tag(ctx)
        d = h.shape[-1]
        h2 = _c(h).view(-1, d)
        tgt = _c(targets).view(-1)
        row_lse, acc = K.lmhead_ce_fwd(h2, emb16_pad, tgt, V, ignore_index=ignore_index)
        out4 = K.combine_losses(acc.data_ptr(), acc.data_ptr() + 4, None, None, 0.0, 0.0, h.device)
        ctx.misc = (emb16_pad, egrad, V, ignore_index, h2.shape[0], d, h.shape)
        ctx.save_for_backward(h2, tgt, row_lse, acc)
        ctx.mark_non_differentiable(acc)
        ddp.expect(ge and any(ctx.needs_input_grad), egrad)
        return out4[1], acc

    @_bw
    def backward(ctx, g, _gacc):
        h2, tgt, row_lse, acc = ctx.saved_tensors
        emb16_pad, egrad, V, ignore_index, R, d, hshape = ctx.misc
        rowp = K.lmhead_ce_rowp(row_lse, tgt, acc, grad_out=_c(g), grad_scale=1.0, ignore_index=ignore_index)
        CH = min(LMHEAD_CHUNK, (V + 7) // 8 * 8)
        # dE (the tied matrix's weight gradient) is needed only by AdamW / the reducer: with side streams on it runs on the
        # weight-gradient stream beside the next chunk's dlogits / dh GEMMs, out of the decoder phase's chain (the least busy part of
        # the step).  Two dlogits buffers alternate; a chunk's buffer is reused only after the side stream has read it.
        side = streams.wgrad_stream() if (egrad is not None and streams.explicit()) else None
        dls = [torch.empty((R, CH), device=h2.device, dtype=BF16) for _ in range(2 if side is not None else 1)]
        dh32 = K.zero_(torch.empty((R, d), device=h2.device, dtype=torch.float32))
        for ci, c0 in enumerate(range(0, V, CH)):
            n = min(CH, V - c0)
            n8 = (n + 7) // 8 * 8                                  # dlogits pad columns are written as zeros; E's pad rows are zero
            dl = dls[ci % len(dls)]
            if side is not None and ci >= 2:
                K.fence(streams.wgrad_raw(), K._stream())          # the dE GEMM of chunk ci - 2 has read this buffer
            K.lmhead_ce_dlogits(h2, emb16_pad, tgt, V, rowp, dl, c0, n, ignore_index=ignore_index)
            ec = emb16_pad[c0:c0 + n8]
            # dh += dlogits_c . E_c: a reduction 8-16x longer than the output is wide -> split-K, fp32 accumulate
            if LMHEAD_FIXUP:
                K.gemm(dl, ec, R, d, n8, out=dh32, ldx=CH, w_kstrided=True, out_mode=2, split_k=4, fixup=True, tile_hint=128)
            else:
                K.gemm(dl, ec, R, d, n8, out=dh32, ldx=CH, w_kstrided=True, out_mode=2, split_k=8)
            if egrad is not None:                                  # dE[c0:c0+n] += dlogits_c^T h
                tiles = ((n + 127) // 128) * ((d + 127) // 128)
                if side is not None:
                    K.fence(K._stream(), streams.wgrad_raw())      # dlogits of this chunk (and h) are ready
                    with K.launch_on(streams.wgrad_raw(), fence=False):
                        K.gemm(dl, h2, n, d, R, out=egrad[c0:c0 + n], ldx=CH, ldw=d, ldo=d, x_kstrided=True, w_kstrided=True, out_mode=2,
                               split_k=K.wgrad_split(R, tiles))
                else:
                    K.gemm(dl, h2, n, d, R, out=egrad[c0:c0 + n], ldx=CH, ldw=d, ldo=d, x_kstrided=True, w_kstrided=True, out_mode=2,
                           split_k=K.wgrad_split(R, tiles))
        dh = K.cast_f32_bf16(dh32)
        ddp.done(egrad)
        return dh.view(hshape), None, None, None, None, None, None, None


def lm_head_ce(h, anchor, emb16_pad, egrad, targets, V, ignore_index=1):
    loss, acc = LmHeadCeFn.apply(h, anchor, emb16_pad, egrad, targets, V, ignore_index, torch.is_grad_enabled())
    return loss, acc


class ColamFn(Function):
    """weight * CoLaM margin loss (TRAIN:296-307,820); gradient flows only into the student states."""

    @staticmethod
    def forward(ctx, hs, hg, mask_u8, margin, weight):
        _tag(ctx)
        loss, cos, ps, pg = K.colam_fwd(_c(hs), _c(hg), mask_u8, margin)
        ctx.misc = (mask_u8, tuple(hs.shape), margin, weight)
        ctx.save_for_backward(cos, ps, pg)
        return loss

    @_bw
    def backward(ctx, g):
        cos, ps, pg = ctx.saved_tensors
        mask_u8, shape, margin, weight = ctx.misc
        return K.colam_bwd(cos, ps, pg, mask_u8, shape, margin, _c(g), weight), None, None, None, None


class SeclaFn(Function):
    """SECLA BatchSoftmax(face, names) (TRAIN:631-660); names carry no gradient (TRAIN:117 no_grad)."""

    @staticmethod
    def forward(ctx, faces, names, weight):
        _tag(ctx)
        faces = _c(faces)
        loss, sim, l1, l2 = K.secla_fwd(faces, names)
        ctx.weight = weight
        ctx.save_for_backward(faces, names, sim, l1, l2)
        return loss

    @_bw
    def backward(ctx, g):
        faces, names, sim, l1, l2 = ctx.saved_tensors
        return K.secla_bwd(faces, names, sim, l1, l2, _c(g), ctx.weight), None, None


_ONES = {}


def const_one(device):
    """fp32 scalar 1.0 on `device`, created once (no fill kernel per step)."""
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones((), device=device, dtype=torch.float32)
    return t


class TotalLossFn(Function):
    """loss = txt + w_secla*secla + w_colam*colam (TRAIN:363).  The weights are applied here in the
    forward value and handed to each loss's backward as grad_scale (ColamFn/SeclaFn `weight`)."""

    @staticmethod
    def forward(ctx, txt, secla, colam, w_secla, w_colam):
        _tag(ctx)
        one = const_one(txt.device)
        out4 = K.combine_losses(txt.data_ptr(), one.data_ptr(), secla, colam, w_secla, w_colam, txt.device)
        ctx.has = (secla is not None, colam is not None)
        ctx.mark_non_differentiable(out4)
        return out4[0], out4

    @_bw
    def backward(ctx, g, _g4):
        return g, (g if ctx.has[0] else None), (g if ctx.has[1] else None), None, None


def total_loss(txt, secla, colam, w_secla, w_colam):
    return TotalLossFn.apply(txt, secla, colam, w_secla, w_colam)
