"""Side HIP streams that fill the GPU's bubbles.

Measured on MI355X (profiles/r1_gemm_overhead.txt): a 256x256-tile GEMM launch carries ~10 us of launch + prologue +
epilogue time plus its store tail, during which most CUs idle, and the chain of ~3000 dependent launches of one training
step exposes all of it.  Two pieces of work are independent of the main chain and are therefore issued on their own
streams so the hardware can schedule their workgroups into those bubbles:
  * the frozen guide-BART forward (TRAIN:293-294) — needed only by the CoLaM loss at the end of the forward — and the
    frozen CLIP ViT forward; both depend only on the batch, NOT on the weights AdamW is still updating, so when the
    caller vouches for the batch (`ready` event) they start while the previous step's AdamW is running;
  * every weight-gradient GEMM + bias-gradient reduction of the backward pass — needed only by AdamW / the DDP reducer.
Ordering is by events.  Tensors consumed on the weight-gradient stream are kept alive by `keep()` until the compute
stream has joined that stream (`join_all`), NOT by Tensor.record_stream(): with recorded blocks outstanding the caching
allocator polls their events on every allocation, which cost ~14 us per torch.empty (16 ms of host time per step).
"""
import os

import torch

from . import kernels as _K

_state = {"enabled": False, "wgrad": None, "aux": None, "vit": None, "branch": None, "wgrad_raw": None, "keep": [],
          "explicit": os.environ.get("VACNIC_EXPLICIT_STREAMS", "1") != "0"}
_K._KEEP = _state["keep"]


def enable(flag=True):
    _state["enabled"] = bool(flag) and torch.cuda.is_available()
    if _state["enabled"] and _state["wgrad"] is None:
        # VACNIC_SIDE_PRIORITY=<int>: HIP priority of the side streams (A/B aid; default = torch's default priority)
        prio = os.environ.get("VACNIC_SIDE_PRIORITY")
        mk = (lambda: torch.cuda.Stream(priority=int(prio))) if prio is not None else torch.cuda.Stream
        _state["wgrad"] = mk()
        _state["wgrad_raw"] = _state["wgrad"].cuda_stream
        _state["aux"] = mk()
        # the two frozen towers share ONE stream (ViT first: the student needs it first): same-box A/B 65.1 vs 66.0 ms/step
        # (profiles/r3_step_ab_towers.txt) — two towers running beside each other AND beside the backward chain take more CUs away
        # from that chain than their overlap buys.  VACNIC_TWO_TOWER_STREAMS=1: one stream each (rounds 1-2).
        # (the torch-context schedule of VACNIC_EXPLICIT_STREAMS=0 enqueues the guide before the ViT: it keeps two streams)
        _state["vit"] = mk() if (os.environ.get("VACNIC_TWO_TOWER_STREAMS") == "1" or not _state["explicit"]) else _state["aux"]
        _state["branch"] = mk()


def explicit():
    """Explicit scheduling (default): side-stream work is launched through kernels.launch_on(raw stream) and every cross-stream
    edge is a kernels.fence() — torch (allocator, autograd engine) sees ONE stream, so no synchronisation is hidden inside
    torch and the step's launch sequence, fences included, can be recorded into a launch plan (training.PlannedTrainStep).
    VACNIC_EXPLICIT_STREAMS=0: torch.cuda.stream() contexts and the autograd engine's own stream syncs (the round-1/2 schedule)."""
    return _state["explicit"]


def raw(name):
    """raw hipStream_t of side stream 'wgrad' | 'aux' | 'vit' | 'branch' (None when side streams are off)."""
    s = _state[name] if _state["enabled"] else None
    return s.cuda_stream if s is not None else None


def enabled():
    return _state["enabled"]


_NO_WGRAD = os.environ.get("VACNIC_NO_WGRAD_STREAM") == "1"       # A/B aids: that work stays on the compute stream
_NO_BRANCH = os.environ.get("VACNIC_NO_BRANCH_STREAM") == "1"


def wgrad_stream():
    return _state["wgrad"] if _state["enabled"] and not _NO_WGRAD else None


def wgrad_raw():
    return _state["wgrad_raw"]


def keep(*tensors):
    """hold references to tensors a side stream is still reading; released by join_all().  Bounded: past 8192 entries the
    compute stream joins the weight-gradient stream early."""
    k = _state["keep"]
    k.extend(tensors)
    if _state["explicit"]:
        if len(k) > (1 << 18):                    # (only a caller that never joins gets here)
            join_all()
        return
    if len(k) > 8192:
        # early release: every stream a kept tensor may have been allocated on (the compute stream and the branch stream, whose
        # blocks go back to THEIR pools) must be ordered behind the weight-gradient stream's reads before the references drop
        torch.cuda.current_stream().wait_stream(_state["wgrad"])
        if _state["branch"] is not None:
            _state["branch"].wait_stream(_state["wgrad"])
        k.clear()


def pending_keep():
    """number of tensors still held for the side streams (0 right after join_all)."""
    return len(_state["keep"])


def branch_stream():
    """stream of the encoder layer's small-token branches (image / face / name streams of MFULL:647-691: a dozen GEMMs over
    20-80 tokens per sample that occupy a fraction of the GPU) — they run beside the text self-attention block of the same
    layer, forward and backward (the autograd engine replays each node on the stream of its forward)."""
    return _state["branch"] if _state["enabled"] and not _NO_BRANCH else None


def side_streams():
    return [s for s in (_state["wgrad"], _state["aux"], _state["vit"], _state["branch"]) if s is not None] if _state["enabled"] else []


def aux_stream():
    return _state["aux"] if _state["enabled"] else None


def vit_stream():
    return _state["vit"] if _state["enabled"] else None


def release_keep():
    """drop the references held for the side streams (the caller has ordered the compute stream behind all of them)."""
    _state["keep"].clear()


def join_all(skip_wgrad=False):
    """make the current stream wait for everything issued on the side streams (before AdamW / the all-reduce tail).
    skip_wgrad: the weight-gradient stream is joined piecewise by the data-parallel reducer (one named event per gradient bucket,
    the last of them behind everything that stream was given); the keep-list then stays until release_keep()."""
    if _state["enabled"] and _state["explicit"]:
        cur = _K._stream()
        for name in ("wgrad", "aux", "vit", "branch"):
            if name == "wgrad" and skip_wgrad:
                continue
            if name != "vit" or _state["vit"] is not _state["aux"]:
                _K.fence(_state[name].cuda_stream, cur)
        if not skip_wgrad:
            _state["keep"].clear()
        return
    if _state["enabled"]:
        cur = torch.cuda.current_stream()
        cur.wait_stream(_state["wgrad"])
        cur.wait_stream(_state["aux"])
        cur.wait_stream(_state["vit"])
        cur.wait_stream(_state["branch"])
        _state["keep"].clear()          # everything the side streams read is now ordered before later compute-stream work
