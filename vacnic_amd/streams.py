"""Side HIP streams that fill the GPU's bubbles.

Measured on MI355X (profiles/r1_gemm_overhead.txt): a 256x256-tile GEMM launch carries ~10 us of launch + prologue +
epilogue time plus its store tail, during which most CUs idle, and the chain of ~3000 dependent launches of one training
step exposes all of it.  Two pieces of work are independent of the main chain and are therefore issued on their own
streams so the hardware can schedule their workgroups into those bubbles:
  * the frozen guide-BART forward (TRAIN:293-294) — needed only by the CoLaM loss at the end of the forward — and the
    frozen CLIP ViT forward; both depend only on the batch, NOT on the weights AdamW is still updating, so when the
    caller vouches for the batch (`ready` event) they start while the previous step's AdamW is running;
  * every weight-gradient GEMM + bias-gradient reduction of the backward pass — needed only by AdamW / the DDP reducer.
Ordering is by events; tensors that cross streams are handed to the caching allocator with record_stream().
"""
import torch

_state = {"enabled": False, "wgrad": None, "aux": None, "vit": None}


def enable(flag=True):
    _state["enabled"] = bool(flag) and torch.cuda.is_available()
    if _state["enabled"] and _state["wgrad"] is None:
        _state["wgrad"] = torch.cuda.Stream()
        _state["aux"] = torch.cuda.Stream()
        _state["vit"] = torch.cuda.Stream()


def enabled():
    return _state["enabled"]


def wgrad_stream():
    return _state["wgrad"] if _state["enabled"] else None


def aux_stream():
    return _state["aux"] if _state["enabled"] else None


def vit_stream():
    return _state["vit"] if _state["enabled"] else None


def join_all():
    """make the current stream wait for everything issued on the side streams (before AdamW / the all-reduce tail)."""
    if _state["enabled"]:
        cur = torch.cuda.current_stream()
        cur.wait_stream(_state["wgrad"])
        cur.wait_stream(_state["aux"])
        cur.wait_stream(_state["vit"])
