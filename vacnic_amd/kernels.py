"""Thin, non-autograd wrappers: torch tensors in, raw device pointers through the C-ABI.

PyTorch here is plumbing only (device memory + the current HIP stream); every arithmetic op on the
path is a hand-written gfx950 kernel in libvacnic_hip.so.  All wrappers launch on
`torch.cuda.current_stream()` so they are hipGraph-capturable, and all fail loudly on CPU tensors.
"""
import contextlib

import torch

from . import _lib
from ._lib import call, call_struct

ACT = {None: 0, "none": 0, "gelu": 1, "tanh": 2, "quick_gelu": 3}
BF16 = torch.bfloat16


_HAVE_GPU = None
_OVERRIDE = None          # raw hipStream_t set by launch_on(): lets ops issue a few launches on a side stream without
                          # paying for torch.cuda.stream()'s context switch (~10 us of host time per use)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw hipStream_t of torch's current stream on the current device (the ~0.3 us C query, not the Python
    `torch.cuda.current_stream()` object that costs ~8 us per call and was 10 ms of host time per training step)."""
    global _HAVE_GPU
    if _HAVE_GPU is None:
        _HAVE_GPU = torch.cuda.is_available()
    if not _HAVE_GPU:
        raise RuntimeError("vacnic_amd kernels need an MI355X (HIP device): there is no CPU fallback")
    if _OVERRIDE is not None:
        return _OVERRIDE
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class launch_on:
    """with launch_on(raw_stream): every vacnic kernel wrapper inside launches on that HIP stream (torch's notion of the
    current stream is untouched, so allocations stay on the caller's stream — the caller owns the ordering/lifetime).

    fence=True (default): the side stream first waits for everything enqueued on torch's current stream so far.  torch's
    allocator believes every block lives on the current stream and hands a freed block out again at once; a kernel launched on
    the side stream into such a block must not overtake the compute-stream kernels that were still using it.  (Blocks a
    side-stream kernel uses are held by the keep-list until the compute stream has joined that stream: kernels._p.)"""

    def __init__(self, raw, fence=True):
        self.raw, self.fence = raw, fence

    def __enter__(self):
        global _OVERRIDE
        self.prev = _OVERRIDE
        if self.fence and self.raw is not None and self.raw != self.prev:
            cur = _raw_stream(torch.cuda.current_device()) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
            if cur != self.raw:
                call("vacnic_stream_fence", cur, self.raw)
        _OVERRIDE = self.raw

    def __exit__(self, *exc):
        global _OVERRIDE
        _OVERRIDE = self.prev
        return False


_KEEP = None            # streams.py's keep-list: tensors handed to a kernel on a side stream stay referenced until the join


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vacnic_amd kernels need CUDA/HIP tensors (there is no CPU fallback)")
    if _OVERRIDE is not None and _KEEP is not None:
        # launched on a side stream while torch's allocator believes everything lives on the current stream: the block must not be
        # handed out again before the compute stream has joined that side stream (streams.join_all drops the references)
        _KEEP.append(t)
    return t.data_ptr()


# ------------------------------------------------------------------------------------- communication
def comm_load():
    """resolve librccl for the comm entry points (the copy torch ships and has already loaded, else the system one)."""
    import os
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    call("vacnic_comm_load", p.encode() if os.path.exists(p) else None)


def comm_unique_id():
    buf = _lib.C.create_string_buffer(128)
    call("vacnic_comm_unique_id", buf)
    return buf.raw


def comm_init(id128, rank, world):
    h = int(_lib.lib.vacnic_comm_init(_lib.C.c_char_p(id128), rank, world))
    if h < 0:
        _lib.check(4)
    return h


def allreduce_bucket(comm, t):
    """in-place SUM all-reduce of a contiguous fp32 / bf16 tensor over the RCCL communicator, on the launch stream."""
    assert t.is_contiguous() and t.dtype in (torch.float32, BF16)
    call("vacnic_allreduce_bucket", comm, _p(t), t.numel(), 0 if t.dtype == torch.float32 else 1, _stream())


def comm_broadcast(comm, t, root=0):
    assert t.is_contiguous() and t.dtype in (torch.float32, BF16)
    call("vacnic_comm_broadcast", comm, _p(t), t.numel(), 0 if t.dtype == torch.float32 else 1, root, _stream())


def event_record(slot, stream=None):
    call("vacnic_event_record", slot, stream if stream is not None else _stream())


def event_wait(slot, stream=None):
    call("vacnic_event_wait", slot, stream if stream is not None else _stream())


def recording():
    """a launch plan is being recorded by this process (vacnic_plan_begin .. vacnic_plan_end)."""
    return int(_lib.lib.vacnic_plan_mark()) >= 0


def fence(src_raw, dst_raw):
    """stream `dst` waits for everything enqueued on stream `src` so far (raw hipStream_t handles) — the recordable form of
    event.record(src); dst.wait_event(event) (vacnic_stream_fence): cross-stream edges are part of a launch plan."""
    if src_raw != dst_raw:
        call("vacnic_stream_fence", src_raw, dst_raw)


def _row_stride(t):
    """t is [..., rows, cols] with unit inner stride and uniformly strided rows when flattened."""
    assert t.stride(-1) == 1, "inner dimension must be contiguous"
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_LOG = None           # tools/autotune_gemm.py sets this to a list to record call signatures
GEMM_TUNED = {}           # "xk,wk,M,N,K,out_mode,has_preact" -> [tile_hint, split_k, ...] measured winners (gemm_tuned.json)


def _load_tuned():
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuned.json")
    if os.path.exists(path) and os.environ.get("VACNIC_GEMM_TUNED", "1") != "0":
        with open(path) as f:
            GEMM_TUNED.update(json.load(f))


_load_tuned()


_SMALLM_SPLIT = __import__("os").environ.get("VACNIC_SMALLM_SPLIT", "0") != "0"      # 1: K slices for the batch-1 fc2 GEMMs of the encoder / ViT (off: see DESIGN section 0, item 5 (c))
_TUNED_FIXUP = __import__("os").environ.get("VACNIC_GEMM_FIXUP_TUNED", "1") != "0"      # A/B: 0 = ignore the fix-up entries of gemm_tuned.json
_FIX = {}                 # launch stream -> [workspace (uint8), counters (int32, all zero between launches)]
_FIX_CAPTURE = {}         # the same for launches recorded by a hipGraph capture on that stream (buffers from the graph's private pool)
_FIX_CAPTURE_FLOOR = 0    # bytes (a debugging aid: tools/enc_graph_check.py)
_CAPTURE_SCOPE = 0        # owner of the graph being captured (capture_scope); 0: graphs that replay in order on one stream
_FIX_OLD = []             # outgrown workspaces: kernels already enqueued may still use them


@contextlib.contextmanager
def capture_scope(tag):
    """fix-up buffers of launches captured inside this block are private to `tag` (any hashable: the object that owns the graphs).
    torch captures every graph on ONE capture stream, so the per-stream table alone would hand the image-tower graph, the encoder
    graph and a decode session's graphs the same workspace and counters — fine while graphs replay one after the other on one stream,
    wrong once generate.CaptionPipeline replays them on several streams at once."""
    global _CAPTURE_SCOPE
    prev, _CAPTURE_SCOPE = _CAPTURE_SCOPE, tag
    try:
        yield
    finally:
        _CAPTURE_SCOPE = prev


def _fix_buffers(stream, M, N, split_k):
    """split-K fix-up buffers of the launch stream: launches on ONE stream run in order, so they share a workspace that only
    grows (persistent: the same addresses at every replay of a launch plan), and the arrival counters, which every launch leaves
    zeroed.  Kept out of torch's per-step allocations on purpose.  The counters of a new buffer are cleared ON THE LAUNCH STREAM
    (vacnic_zero_bytes: ordered before the GEMM whatever stream torch considers current; a fill kernel inside a stream capture).
    Launches recorded by a hipGraph capture get buffers of their own (allocated from the graph's pool): a captured launch and an
    eager one must never share counters; graphs captured in one scope (capture_scope) replay in order on one stream."""
    need = int(_lib.lib.vacnic_gemm_workspace_bytes(M, N, split_k))
    ncnt = int(_lib.lib.vacnic_gemm_counters(M, N))
    capturing = torch.cuda.is_current_stream_capturing()
    table = _FIX_CAPTURE if capturing else _FIX
    # graphs that may replay CONCURRENTLY (generate.GraphedCall: image tower / encoder) never share buffers
    key = (stream, _CAPTURE_SCOPE) if capturing else stream
    ent = table.get(key)
    if ent is None or ent[0].numel() < need or ent[1].numel() < ncnt:
        floor = _FIX_CAPTURE_FLOOR if table is _FIX_CAPTURE else 64 << 20
        ws = torch.empty(max(need, ent[0].numel() if ent else 0, floor), device="cuda", dtype=torch.uint8)
        cnt = torch.empty(max(ncnt, ent[1].numel() if ent else 0, 4096), device="cuda", dtype=torch.int32)
        call("vacnic_zero_bytes", cnt.data_ptr(), cnt.numel() * 4, stream)
        if ent is not None:
            _FIX_OLD.append(ent)
        ent = table[key] = [ws, cnt]
    return ent


def gemm(x, w, M, N, K, *, bias=None, out=None, ldx=None, ldw=None, ldo=None, x_kstrided=False, w_kstrided=False,
         act=None, out_mode=0, split_k=1, alpha=1.0, preact=None, dact_src=None, residual=None, tile_hint=0, xsum=None, fixup=False,
         drop=None):
    """out[m][n] = epi(alpha * sum_k X(m,k) W(n,k) + bias[n]) — see include/vacnic_hip.h.
    tile_hint 0: measured winner for this exact shape if gemm_tuned.json has one, else the C-side cost model; -1: cost model.
    fixup: split_k > 1 through the ordered fix-up (no atomics, bitwise reproducible, any out_mode) instead of fp32 atomics.
    drop: (p, seed, seed_dev tensor or None) — activation dropout in the epilogue (after act / act'); see can_fuse_dropout."""
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=BF16 if out_mode == 0 else torch.float32)
    ldx = ldx if ldx is not None else (M if x_kstrided else K)
    ldw = ldw if ldw is not None else (N if w_kstrided else K)
    ldo = ldo if ldo is not None else N
    if GEMM_LOG is not None:
        GEMM_LOG.append((int(x_kstrided), int(w_kstrided), M, N, K, ldx, ldw, ldo, out_mode, split_k, bias is not None, act,
                         preact is not None, dact_src is not None, residual is not None))
    if tile_hint == 0 and GEMM_TUNED:
        t = GEMM_TUNED.get(f"{int(x_kstrided)},{int(w_kstrided)},{M},{N},{K},{out_mode},{int(preact is not None)}")
        if t is not None and len(t) > 4 and t[4]:
            # measured winner = K slices through the ordered fix-up (long reductions into a small output: too few tiles to fill
            # 256 CUs unsplit, and a bf16 / activation / residual epilogue cannot meet in atomics)
            if _TUNED_FIXUP and split_k == 1:
                tile_hint, split_k, fixup = t[0], t[1], True
        elif t is not None:
            tile_hint = t[0] or -1
            if out_mode == 2:
                split_k = t[1] if _SPLIT_SCALE == 1.0 else max(1, min(32, int(t[1] * _SPLIT_SCALE)))
        elif (t is None and _SMALLM_SPLIT and split_k == 1 and not fixup and 8 < M <= 1024 and K >= 4096 and N <= 2048 and out_mode != 2
              and xsum is None and not x_kstrided and not w_kstrided and not torch.is_grad_enabled()):
            # one caption's encoder / ViT pass (batch 1: M = 512 tokens, 257 patches), fc2: 64 x 128 tiles give <= 64 workgroups, each
            # alone on its CU with a 64-iteration K loop bound by the latency of its own loads (30 us); four K slices through the
            # ordered fix-up: 17-18 us (profiles/r4_small_m_gemm.txt).  K = 1024 launches sit on an 11 us floor either way.
            tile_hint, split_k, fixup = 64, 4, True
    if tile_hint < 0:
        tile_hint = 0
    st = _stream()
    ws = wsb = cnt = cntn = None
    if fixup and split_k > 1:
        wbuf, cbuf = _fix_buffers(st, M, N, split_k)
        ws, wsb, cnt, cntn = wbuf.data_ptr(), wbuf.numel(), cbuf.data_ptr(), cbuf.numel()
    call_struct("vacnic_gemm_bf16", stream=st, x=_p(x), w=_p(w), bias=_p(bias), out=_p(out), preact=_p(preact),
                dact_src=_p(dact_src), residual=_p(residual), xsum=_p(xsum), M=M, N=N, K=K, ldx=ldx, ldw=ldw, ldo=ldo,
                x_kstrided=int(x_kstrided), w_kstrided=int(w_kstrided), act=ACT[act], out_mode=out_mode,
                split_k=split_k, alpha=alpha, tile_hint=tile_hint, workspace=ws, workspace_bytes=wsb or 0, counters=cnt,
                counters_len=cntn or 0, drop_p=float(drop[0]) if drop else 0.0, drop_seed=int(drop[1]) if drop else 0,
                drop_seed_dev=_p(drop[2]) if drop else None)
    return out


def can_fuse_dropout(N, ldo=None):
    """the GEMM epilogue can apply activation dropout to a contiguous [M, N] bf16 output with N % 16 == 0."""
    return N % 16 == 0 and (ldo is None or ldo == N)


def gemv_ln(x, residual, gamma, beta, w, M, N, Kd, *, bias=None, out=None, ln_out=None, ldw=None, ldo=None, act=None, out_mode=0, eps=1e-5):
    """out = epi(LayerNorm(x + residual) . w^T + bias) for M <= 8 rows; ln_out (optional) receives the normalised rows."""
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=BF16 if out_mode == 0 else torch.float32)
    call_struct("vacnic_gemv_ln_bf16", stream=_stream(), x=_p(x), residual=_p(residual), gamma=_p(gamma), beta=_p(beta), ln_out=_p(ln_out),
                w=_p(w), bias=_p(bias), out=_p(out), M=M, N=N, K=Kd, ldw=ldw if ldw is not None else Kd, ldo=ldo if ldo is not None else N,
                act=ACT[act], out_mode=out_mode, eps=eps)
    return out


_SPLIT_SCALE = float(__import__("os").environ.get("VACNIC_WGRAD_SPLIT_SCALE", "1"))     # A/B aid: finer / coarser K slices in the step


def wgrad_split(M_red, n_tiles):
    """split-K factor for a weight gradient whose reduction runs over M_red rows: aim at >= 512 workgroups."""
    s = 1
    while n_tiles * s < 512 and M_red // (s * 2) >= 512 and s < 16:
        s *= 2
    if _SPLIT_SCALE != 1.0:
        s = max(1, min(32, int(s * _SPLIT_SCALE)))
        while s > 1 and M_red // s < 256:
            s //= 2
    return s


def wgrad_group(jobs):
    """jobs: list of (dy2d [M,N] bf16, x2d [M,K] bf16, dw [N,K] f32 view, dbias [N] f32 or None): dw += dy^T x, dbias += colsum(dy)
    for all of them in one launch per 16 output blocks of 1024 x 1024 — no split-K, no atomics on dw (vacnic_wgrad_group)."""
    arr = (_lib.WgradJob * len(jobs))()
    for j, (dy, x, dw, db) in enumerate(jobs):
        M, N = dy.shape
        Kd = x.shape[1]
        assert x.shape[0] == M and tuple(dw.shape) == (N, Kd) and dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1
        a = arr[j]
        a.dy, a.x, a.dw, a.dbias = _p(dy), _p(x), _p(dw), _p(db)
        a.M, a.N, a.K, a.lddy, a.ldx, a.lddw = M, N, Kd, dy.stride(0), x.stride(0), dw.stride(0)
    call("vacnic_wgrad_group", arr, len(jobs), _stream())


# ------------------------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=None, causal=False, scale=0.125, need_lse=True, p_drop=0.0, seed=0, seed_dev=None):
    """q/k/v: [B, T, >=H*64] views (unit inner stride); returns out [B,Tq,H*64] bf16 and lse [B,H,Tq]."""
    out = torch.empty((B, Tq, H * 64), device=q.device, dtype=BF16)
    lse = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32) if need_lse else None
    call_struct("vacnic_attn_fwd", stream=_stream(), q=_p(q), k=_p(k), v=_p(v), out=_p(out), lse=_p(lse),
                key_mask=_p(key_mask), B=B, H=H, Tq=Tq, Tk=Tk, ldq=q.stride(1), ldk=k.stride(1), ldv=v.stride(1),
                ldo=out.stride(1), bsq=q.stride(0), bsk=k.stride(0), bsv=v.stride(0), bso=out.stride(0),
                causal=int(causal), scale=scale, p_drop=p_drop, seed=seed, seed_dev=_p(seed_dev))
    return out, lse


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, key_mask=None, causal=False, scale=0.125, p_drop=0.0, seed=0, seed_dev=None):
    delta = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    assert dout.stride() == out.stride()
    call_struct("vacnic_attn_bwd", stream=_stream(), q=_p(q), k=_p(k), v=_p(v), out=_p(out), dout=_p(dout), lse=_p(lse),
                delta=_p(delta), dq=_p(dq), dk=_p(dk), dv=_p(dv), key_mask=_p(key_mask), B=B, H=H, Tq=Tq, Tk=Tk,
                ldq=q.stride(1), ldk=k.stride(1), ldv=v.stride(1), ldo=out.stride(1),
                bsq=q.stride(0), bsk=k.stride(0), bsv=v.stride(0), bso=out.stride(0),
                lddq=dq.stride(1), lddk=dk.stride(1), lddv=dv.stride(1),
                bsdq=dq.stride(0), bsdk=dk.stride(0), bsdv=dv.stride(0), causal=int(causal), scale=scale, p_drop=p_drop, seed=seed,
                seed_dev=_p(seed_dev))


# -------------------------------------------------------------------------------------------- LN family
LN_TWO_STAGE = __import__("os").environ.get("VACNIC_LN_ATOMICS") != "1"     # A/B: VACNIC_LN_ATOMICS=1 = round-1 per-column atomics

def add_ln_fwd(x, residual, gamma, beta, eps=1e-5, p_drop=0.0, seed=0, need_stats=True, seed_dev=None):
    D = x.shape[-1]
    R = x.numel() // D
    out = torch.empty_like(x)
    mean = torch.empty(R, device=x.device, dtype=torch.float32) if need_stats else None
    rstd = torch.empty(R, device=x.device, dtype=torch.float32) if need_stats else None
    call_struct("vacnic_add_ln_fwd", stream=_stream(), x=_p(x), residual=_p(residual), gamma=_p(gamma), beta=_p(beta),
                out=_p(out), mean=_p(mean), rstd=_p(rstd), R=R, D=D, eps=eps, p_drop=p_drop, seed=seed, seed_dev=_p(seed_dev))
    return out, mean, rstd


def add_ln_bwd(dout, x, residual, gamma, mean, rstd, dgamma, dbeta, p_drop=0.0, seed=0, need_dres=True, seed_dev=None, fold_on=None):
    """fold_on: raw stream for the dgamma / dbeta fold of the two-stage reduction (None: in the call, on the launch stream).  The
    fold is a 5 us kernel nobody in the backward chain waits for; the caller orders `fold_on` behind this launch's stream."""
    D = x.shape[-1]
    R = x.numel() // D
    dx = torch.empty_like(x)
    if residual is None or not need_dres:
        dres = None
    elif p_drop > 0.0:
        dres = torch.empty_like(x)
    else:
        dres = None            # identical to dx: caller reuses dx
    part, prow = None, 0
    if LN_TWO_STAGE and dgamma is not None and R >= 256:
        prow = min(1024, (R + 3) // 4)
        part = torch.empty((prow, 2, D), device=x.device, dtype=torch.float32)     # scratch of the two-stage dgamma/dbeta fold
    defer = fold_on is not None and part is not None and fold_on != _stream()
    call_struct("vacnic_add_ln_bwd", stream=_stream(), dout=_p(dout), x=_p(x), residual=_p(residual), gamma=_p(gamma),
                mean=_p(mean), rstd=_p(rstd), dresidual=_p(dres), dx=_p(dx), dgamma=_p(dgamma), dbeta=_p(dbeta),
                R=R, D=D, p_drop=p_drop, seed=seed, seed_dev=_p(seed_dev), partials=_p(part), partial_rows=prow, defer_fold=int(defer))
    if defer:
        fence(_stream(), fold_on)
        with launch_on(fold_on, fence=False):
            call("vacnic_ln_partial_fold", _p(part), _p(dgamma), _p(dbeta), prow, D, _stream())
    return dx, (dres if dres is not None else dx)


def dropout_(x, p_drop, seed, seed_dev=None):
    """in-place inverted dropout of a contiguous bf16 tensor (activation dropout; the same call on the gradient is its backward)."""
    assert x.is_contiguous() and x.dtype == BF16
    call("vacnic_dropout_bf16", _p(x), _p(x), x.numel(), float(p_drop), int(seed), _p(seed_dev), _stream())
    return x


def embed_ln_fwd(ids, embed16, pos16, gamma, beta, embed_scale=1.0, pos_offset=2, eps=1e-5, p_drop=0.0, seed=0, seed_dev=None):
    B, T = ids.shape
    V, D = embed16.shape
    out = torch.empty((B, T, D), device=ids.device, dtype=BF16)
    mean = torch.empty(B * T, device=ids.device, dtype=torch.float32)
    rstd = torch.empty(B * T, device=ids.device, dtype=torch.float32)
    call_struct("vacnic_embed_ln_fwd", stream=_stream(), ids=_p(ids), embed=_p(embed16), pos=_p(pos16), gamma=_p(gamma),
                beta=_p(beta), out=_p(out), mean=_p(mean), rstd=_p(rstd), B=B, T=T, D=D, V=V, pos_offset=pos_offset,
                embed_scale=embed_scale, eps=eps, p_drop=p_drop, seed=seed, seed_dev=_p(seed_dev))
    return out, mean, rstd


def embed_ln_bwd(ids, embed16, pos16, dout, gamma, mean, rstd, dembed, dpos, dgamma, dbeta, embed_scale=1.0,
                 pos_offset=2, padding_idx=1, p_drop=0.0, seed=0, seed_dev=None):
    B, T = ids.shape
    V, D = embed16.shape
    call_struct("vacnic_embed_ln_bwd", stream=_stream(), ids=_p(ids), embed=_p(embed16), pos=_p(pos16), dout=_p(dout),
                gamma=_p(gamma), mean=_p(mean), rstd=_p(rstd), dembed=_p(dembed), dpos=_p(dpos), dgamma=_p(dgamma),
                dbeta=_p(dbeta), B=B, T=T, D=D, V=V, pos_offset=pos_offset, embed_scale=embed_scale,
                padding_idx=padding_idx, p_drop=p_drop, seed=seed, seed_dev=_p(seed_dev))


def name_embed_mean(ids3d, embed16, pos16, gamma, beta, embed_scale=1.0, pos_offset=2, eps=1e-5):
    B, Nn, Ln = ids3d.shape
    V, D = embed16.shape
    out = torch.empty((B, Nn, D), device=ids3d.device, dtype=torch.float32)
    call_struct("vacnic_name_embed_mean", stream=_stream(), ids=_p(ids3d), embed=_p(embed16), pos=_p(pos16),
                gamma=_p(gamma), beta=_p(beta), out=_p(out), B=B, Nn=Nn, Ln=Ln, D=D, V=V, pos_offset=pos_offset,
                embed_scale=embed_scale, eps=eps)
    return out


# ------------------------------------------------------------------------------------------------ losses
def ce_fwd(logits, targets, V, ignore_index=1):
    R, ldl = logits.shape[0], logits.stride(0)
    dev = logits.device
    row_lse = torch.empty(R, device=dev, dtype=torch.float32)
    acc = torch.zeros(2, device=dev, dtype=torch.float32)       # {loss_sum, count}
    call_struct("vacnic_ce_fwd", stream=_stream(), logits=_p(logits), targets=_p(targets), row_lse=_p(row_lse),
                row_loss=None, loss_sum=acc.data_ptr(), count=acc.data_ptr() + 4, dlogits=None, grad_out=None,
                grad_scale=1.0, R=R, V=V, ldl=ldl, ldd=ldl, ignore_index=ignore_index,
                logits_f32=int(logits.dtype == torch.float32))
    return row_lse, acc


def ce_bwd(logits, targets, V, row_lse, acc, dlogits, grad_out=None, grad_scale=1.0, ignore_index=1):
    R, ldl = logits.shape[0], logits.stride(0)
    call_struct("vacnic_ce_bwd", stream=_stream(), logits=_p(logits), targets=_p(targets), row_lse=_p(row_lse),
                row_loss=None, loss_sum=acc.data_ptr(), count=acc.data_ptr() + 4, dlogits=_p(dlogits),
                grad_out=_p(grad_out), grad_scale=grad_scale, R=R, V=V, ldl=ldl, ldd=dlogits.stride(0),
                ignore_index=ignore_index, logits_f32=int(logits.dtype == torch.float32))


def zero_(t):
    """t.zero_() as a memset node on the launch stream (no fill kernel)."""
    assert t.is_contiguous()
    call("vacnic_zero_bytes", _p(t), t.numel() * t.element_size(), _stream())
    return t


def _lmhead_args(h2, emb16, targets, V, ignore_index, part=None, tl=None, row_lse=None, acc=None):
    R, D = h2.shape
    return _lib.LmheadCeArgs(h=_p(h2), emb=_p(emb16), bias=None, targets=_p(targets), part=_p(part), tl=_p(tl), row_lse=_p(row_lse),
                             loss_sum=acc.data_ptr() if acc is not None else None, count=acc.data_ptr() + 4 if acc is not None else None,
                             R=R, V=V, D=D, ldh=h2.stride(0), lde=emb16.stride(0), part_tiles=(V + 255) // 256, ignore_index=ignore_index)


def lmhead_ce_fwd(h2, emb16, targets, V, ignore_index=1):
    """fused lm_head + CrossEntropyLoss forward without logits: returns (row_lse [R], acc = {loss_sum, count})."""
    R = h2.shape[0]
    dev = h2.device
    tiles = (V + 255) // 256
    part = torch.empty((R, tiles, 2), device=dev, dtype=torch.float32)
    tl = torch.empty(R, device=dev, dtype=torch.float32)
    row_lse = torch.empty(R, device=dev, dtype=torch.float32)
    acc = torch.empty(2, device=dev, dtype=torch.float32)
    st = _lmhead_args(h2, emb16, targets, V, ignore_index, part, tl, row_lse, acc)
    _lib.check(_lib.lib.vacnic_lmhead_ce_fwd(_lib.C.byref(st), _stream()))
    return row_lse, acc


def lmhead_ce_rowp(row_lse, targets, acc, grad_out=None, grad_scale=1.0, ignore_index=1):
    R = row_lse.shape[0]
    rowp = torch.empty((R, 2), device=row_lse.device, dtype=torch.float32)
    call("vacnic_lmhead_ce_rowp", _p(row_lse), _p(targets), acc.data_ptr() + 4, _p(grad_out), grad_scale, _p(rowp), R, ignore_index, _stream())
    return rowp


def lmhead_ce_dlogits(h2, emb16, targets, V, rowp, dl, col0, ncols, ignore_index=1):
    """dl[:, :round_up(ncols, 8)] = bf16 dlogits of vocabulary columns [col0, col0 + ncols) (logits recomputed on chip)."""
    st = _lmhead_args(h2, emb16, targets, V, ignore_index)
    _lib.check(_lib.lib.vacnic_lmhead_ce_dlogits(_lib.C.byref(st), col0, ncols, _p(dl), dl.stride(0), _p(rowp), _stream()))


def combine_losses(ce_sum_ptr, count_ptr, secla, colam, w_secla, w_colam, device):
    out4 = torch.empty(4, device=device, dtype=torch.float32)
    call("vacnic_combine_losses", ce_sum_ptr, count_ptr, _p(secla), _p(colam), w_secla, w_colam, out4.data_ptr(), _stream())
    return out4


def colam_fwd(hs, hg, mask_u8, margin):
    B, T, D = hs.shape
    dev = hs.device
    loss = torch.empty((), device=dev, dtype=torch.float32)
    cos = torch.empty(B, device=dev, dtype=torch.float32)
    ps = torch.empty((B, D), device=dev, dtype=torch.float32)
    pg = torch.empty((B, D), device=dev, dtype=torch.float32)
    call_struct("vacnic_colam_fwd", stream=_stream(), hs=_p(hs), hg=_p(hg), mask=_p(mask_u8), loss=_p(loss), cos=_p(cos),
                pooled_s=_p(ps), pooled_g=_p(pg), B=B, T=T, D=D, margin=margin)
    return loss, cos, ps, pg


def colam_bwd(cos, ps, pg, mask_u8, shape, margin, grad_out, grad_scale):
    B, T, D = shape
    dhs = torch.empty(shape, device=cos.device, dtype=BF16)
    call_struct("vacnic_colam_bwd", stream=_stream(), cos=_p(cos), pooled_s=_p(ps), pooled_g=_p(pg), mask=_p(mask_u8),
                dhs=_p(dhs), B=B, T=T, D=D, margin=margin, grad_out=_p(grad_out), grad_scale=grad_scale)
    return dhs


def secla_fwd(faces, names):
    B, F, D = faces.shape
    N = names.shape[1]
    dev = faces.device
    sim = torch.empty((B, N, B, F), device=dev, dtype=torch.float32)
    l1 = torch.empty((B, B), device=dev, dtype=torch.float32)
    l2 = torch.empty((B, B), device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    call_struct("vacnic_secla_fwd", stream=_stream(), faces=_p(faces), names=_p(names), sim=_p(sim), logits1=_p(l1),
                logits2=_p(l2), loss=_p(loss), B=B, F=F, N=N, D=D)
    return loss, sim, l1, l2


def secla_bwd(faces, names, sim, l1, l2, grad_out, grad_scale):
    B, F, D = faces.shape
    N = names.shape[1]
    dfaces = torch.empty((B, F, D), device=faces.device, dtype=BF16)
    wsim = torch.empty_like(sim)
    call_struct("vacnic_secla_bwd", stream=_stream(), faces=_p(faces), names=_p(names), sim=_p(sim), logits1=_p(l1),
                logits2=_p(l2), dfaces=_p(dfaces), wsim=_p(wsim), B=B, F=F, N=N, D=D, grad_out=_p(grad_out),
                grad_scale=grad_scale)
    return dfaces


# ---------------------------------------------------------------------------------------------- optimizer
def lr_step(hyper, base_lr, warmup, total, rng_counter=None):
    call("vacnic_lr_step", hyper.data_ptr(), base_lr, float(warmup), float(total), _p(rng_counter), _stream())


def adamw(p, g, m, v, p16, hyper, n, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0, zero_grad=True,
          clip_coef=None):
    call_struct("vacnic_adamw", stream=_stream(), p=_p(p), g=_p(g), m=_p(m), v=_p(v), p_bf16=_p(p16), hyper=_p(hyper),
                n=n, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale,
                zero_grad=int(zero_grad), clip_coef=_p(clip_coef))


def grad_clip_coef(g, n, max_norm, grad_scale=1.0, partials=None, out=None):
    """clip_grad_norm_ (TRAIN:365-366) on device: returns the fp32 pair {clip coefficient, total norm}."""
    if partials is None:
        partials = torch.empty(1024, device=g.device, dtype=torch.float32)
    if out is None:
        out = torch.empty(2, device=g.device, dtype=torch.float32)
    assert g.dtype == torch.float32 and partials.numel() >= 1024 and out.numel() >= 2
    call("vacnic_grad_clip_coef", _p(g), n, grad_scale, max_norm, _p(partials), _p(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------- misc
def cast_f32_bf16(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    assert src.is_contiguous() and dst.is_contiguous()
    call("vacnic_cast_f32_bf16", _p(src), _p(dst), src.numel(), _stream())
    return dst


def cast_bf16_f32(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    assert src.is_contiguous() and dst.is_contiguous()
    call("vacnic_cast_bf16_f32", _p(src), _p(dst), src.numel(), _stream())
    return dst


def copy3d(src, dst, B, rows, cols, accumulate=False):
    """src/dst: [B, rows, cols] views with unit inner stride."""
    call("vacnic_copy3d_bf16", _p(src), _p(dst), B, rows, cols, src.stride(1), dst.stride(1), src.stride(0), dst.stride(0),
         int(accumulate), _stream())


def zero_strided(t):
    """zero a [B, rows, cols] bf16 view with unit inner stride (rare fallback: an unused slot of a batched gradient buffer)."""
    z = torch.empty((t.shape[0], t.shape[1], t.shape[2]), device=t.device, dtype=t.dtype)
    zero_(z)
    copy3d(z, t, t.shape[0], t.shape[1], t.shape[2])


def cat_tokens(parts):
    """torch.cat(parts, dim=1) for [B, T_i, D] bf16 tensors (MFULL:666,691)."""
    B, _, D = parts[0].shape
    T = sum(p.shape[1] for p in parts)
    out = torch.empty((B, T, D), device=parts[0].device, dtype=BF16)
    o = 0
    for p in parts:
        copy3d(p, out[:, o:o + p.shape[1]], B, p.shape[1], D)
        o += p.shape[1]
    return out


def pad_cols(x2d, Cp):
    """[R, C] -> [R, Cp] zero padded (C not a multiple of 8)."""
    R, C = x2d.shape
    out = torch.empty((R, Cp), device=x2d.device, dtype=BF16)
    call("vacnic_pad_cols_bf16", _p(x2d), _p(out), R, C, Cp, x2d.stride(0), _stream())
    return out


def add(a, b):
    out = torch.empty_like(a)
    assert a.is_contiguous() and b.is_contiguous()
    call("vacnic_add_bf16", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def im2col_patches(img, patch, Kp):
    B, _, HW, _ = img.shape
    g = HW // patch
    out = torch.empty((B * g * g, Kp), device=img.device, dtype=BF16)
    call("vacnic_im2col_patches", _p(img), _p(out), B, HW, patch, Kp, _stream())
    return out


def vit_assemble(patch_emb, cls16, pos16, B, G2, W):
    out = torch.empty((B, G2 + 1, W), device=patch_emb.device, dtype=BF16)
    call("vacnic_vit_assemble", _p(patch_emb), _p(cls16), _p(pos16), _p(out), B, G2, W, _stream())
    return out


def prep_ids(ids, pad_id=1, start_id=None, want_mask=True):
    B, T = ids.shape
    mask = torch.empty((B, T), device=ids.device, dtype=torch.uint8) if want_mask else None
    shifted = torch.empty_like(ids) if start_id is not None else None
    call("vacnic_prep_ids", _p(ids), _p(mask), _p(shifted), B, T, pad_id, start_id if start_id is not None else 0, _stream())
    return mask, shifted


def prep_ids_into(ids, mask, shifted, pad_id=1, start_id=None):
    """prep_ids writing into caller-owned buffers (either may be None)."""
    B, T = ids.shape
    call("vacnic_prep_ids", _p(ids), _p(mask), _p(shifted), B, T, pad_id, start_id if start_id is not None else 0, _stream())


def face_mask(face_emb):
    B, F, D = face_emb.shape
    mask = torch.empty((B, F), device=face_emb.device, dtype=torch.uint8)
    call("vacnic_face_mask", _p(face_emb), _p(mask), B * F, D, _stream())
    return mask


def cat_masks(a, b):
    """torch.cat((a, b), dim=1) for two uint8 [B, n] masks (MFULL:1262)."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and a.is_contiguous() and b.is_contiguous() and a.shape[0] == b.shape[0]
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), device=a.device, dtype=torch.uint8)
    call("vacnic_cat2_u8", _p(a), _p(b), _p(out), a.shape[0], a.shape[1], b.shape[1], _stream())
    return out


def argmax_rows(logits, V):
    R = logits.shape[0]
    out = torch.empty(R, device=logits.device, dtype=torch.int64)
    call("vacnic_argmax_rows", _p(logits), _p(out), R, V, logits.stride(0), int(logits.dtype == torch.float32), _stream())
    return out


def bias_grad(dy2d, dbias, M, N):
    call("vacnic_bias_grad", _p(dy2d), _p(dbias), M, N, dy2d.stride(0), _stream())


def probe_layouts():
    out = torch.zeros(2624, device="cuda", dtype=torch.float32)
    src = torch.arange(128, device="cuda", dtype=torch.float32) + 1.0
    call("vacnic_probe_layouts", _p(out), _p(src), out.numel(), _stream())
    return out


# ------------------------------------------------------------------------------------------------ decode
def beam_topk(logits, V, K_, beam_scores=None, bans=None, eos=2, suppress_eos=False, forced_token=-1):
    R = logits.shape[0]
    tv = torch.empty((R, K_), device=logits.device, dtype=torch.float32)
    ti = torch.empty((R, K_), device=logits.device, dtype=torch.int32)
    call("vacnic_beam_topk", _p(logits), _p(beam_scores), _p(bans), bans.shape[1] if bans is not None else 0, eos, int(suppress_eos),
         forced_token, _p(tv), _p(ti), R, V, logits.stride(0), K_, int(logits.dtype == torch.float32), _stream())
    return tv, ti


def lmhead_topk(h, emb, V, K_, *, bias=None, beam_scores=None, bans=None, eos=2, suppress_eos=False, forced_token=-1, logits=None):
    """(top values, top ids) [R, K_] of log_softmax(h . emb[:V]^T + bias) after the logits processors, + beam_scores — the LM head
    and vacnic_beam_topk of one decode position without the [R, V] logits (R <= 8, d_model <= 1024).  A forced position
    (forced_token >= 0) computes no logits at all.  logits: optional fp32 [R, >= V] buffer that receives them (diagnostics)."""
    R, d = h.shape
    tv = torch.empty((R, K_), device=h.device, dtype=torch.float32)
    ti = torch.empty((R, K_), device=h.device, dtype=torch.int32)
    nws = int(_lib.lib.vacnic_lmhead_topk_workspace(R, V, K_)) if forced_token < 0 else 0
    ws = torch.empty(nws, device=h.device, dtype=torch.float32) if nws else None
    call_struct("vacnic_lmhead_topk", stream=_stream(), h=_p(h), emb=_p(emb), bias=_p(bias), logits=_p(logits) if forced_token < 0 else None,
                workspace=_p(ws), beam_scores=_p(beam_scores), bans=_p(bans), top_val=_p(tv), top_idx=_p(ti), R=R, V=V, d=d, ldw=emb.stride(0),
                ldl=logits.stride(0) if logits is not None else 0, workspace_floats=nws, n_ban=bans.shape[1] if bans is not None else 0, eos=eos,
                suppress_eos=int(suppress_eos), forced_token=forced_token, K2=K_)
    return tv, ti


def image_u8_normalize(img_u8, flip=None, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)):
    """uint8 [B,3,H,W] (+ optional uint8 flip flags [B]) -> fp32 ToTensor + Normalize (TRAIN:741-764), bit-identical to torch."""
    B, C, H, W = img_u8.shape
    assert C == 3 and img_u8.dtype == torch.uint8 and img_u8.is_contiguous()
    out = torch.empty((B, 3, H, W), device=img_u8.device, dtype=torch.float32)
    call("vacnic_image_u8_normalize", _p(img_u8), _p(flip), _p(out), B, H, W, *[float(v) for v in mean], *[float(v) for v in std], _stream())
    return out


def gather_rows(src, dst, idx, rows, row_bytes, row_stride_bytes=0, period=0):
    call("vacnic_gather_rows", _p(src), _p(dst), _p(idx), rows, row_bytes, row_stride_bytes, period, _stream())
