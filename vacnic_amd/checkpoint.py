"""Resumable checkpoints.

The reference pickles the whole module object each epoch (`torch.save(model, ...)`, TRAIN:464-472) and keeps no optimizer,
scheduler or RNG state, so a run can be evaluated but not resumed.  Here a checkpoint is a plain dict of tensors keyed by
the REFERENCE's parameter names (the modules mirror `src/models`, so `state_dict()` of a reference model built from the
same config loads it and vice versa), plus everything the continuation of a run depends on — weights, AdamW moments, schedule position and
the dropout RNG are restored EXACTLY (the first step after a resume reproduces the uninterrupted run to fp32 round-off, 2e-5); later steps
agree to ~1e-3 because a step itself is not run-to-run deterministic: the split-K weight gradients, the LayerNorm parameter gradients and the
embedding scatter sum through fp32 atomics whose order varies (VACNIC_WGRAD_GROUP_MAX_M=1000000 removes the first source):
  model      {reference parameter name: fp32 tensor}   (the fp32 master copies; the bf16 shadow is derived)
  optimizer  {"exp_avg": {name: tensor}, "exp_avg_sq": {name: tensor}, "lr": float, "step": int}   — torch.optim.AdamW layout
  schedule   {"base_lr", "num_warmup_steps", "num_training_steps"}
  rng        {"seed", "counter", "device_counter"}      — Philox dropout seeds (ops.Rng); `seed` is rank-free, rank r uses seed + r
  meta       {"format": 1, "step": int, ...caller extras}
Everything is moved to the CPU before `torch.save`, so a checkpoint loads on any box."""
import torch

from . import ops
from .ddp import DistributedDataParallel

FORMAT = 1


def _net(model):
    return model.module if isinstance(model, DistributedDataParallel) else model


def _named_trainable(net):
    seen = set()
    for name, p in net.named_parameters():
        if name.startswith("clip_model.") or id(p) in seen:
            continue
        seen.add(id(p))
        yield name, p


def model_state(model):
    """reference-named fp32 weights (tied lm_head / decoder embedding listed once, under `model.shared.weight`, plus the
    aliases the reference's state_dict carries)."""
    net = _net(model)
    sd = {}
    for name, t in net.state_dict().items():
        if name.startswith("clip_model."):
            continue
        sd[name] = t.detach().to("cpu", torch.float32).clone()
    return sd


def optimizer_state(model, optimizer):
    net, a = _net(model), optimizer.arena
    ea, es = {}, {}
    for name, p in _named_trainable(net):
        if id(p) not in a.slots:
            continue
        o, n, _ = a.slots[id(p)]
        ea[name] = a.exp_avg[o:o + n].view(p.shape).detach().cpu().clone()
        es[name] = a.exp_avg_sq[o:o + n].view(p.shape).detach().cpu().clone()
    hyper = optimizer.hyper.detach().cpu()
    return {"exp_avg": ea, "exp_avg_sq": es, "lr": float(hyper[0]), "step": int(hyper[1])}


def save_checkpoint(path, model, optimizer=None, step=0, rank=0, **extra):
    """rank: the saving process's data-parallel rank (its dropout seed offset)."""
    ck = {"model": model_state(model), "meta": dict(extra, format=FORMAT, step=int(step))}
    if optimizer is not None:
        ck["optimizer"] = optimizer_state(model, optimizer)
        ck["schedule"] = {"base_lr": optimizer.lr, "num_warmup_steps": optimizer.warmup, "num_training_steps": optimizer.total,
                          "weight_decay": optimizer.wd, "betas": tuple(optimizer.betas), "eps": optimizer.eps}
    dev = ops.Rng.dev
    # `seed` is the run's seed WITHOUT the saving rank's offset (the trainer seeds rank r with seed + r): a resumed rank
    # re-derives its own base, so the data-parallel replicas keep drawing different dropout masks
    ck["rng"] = {"seed": (ops.Rng.base - int(rank)) & 0xFFFFFFFF, "counter": ops.Rng.counter,
                 "device_counter": int(dev.item()) if dev is not None else 0}
    torch.save(ck, path)
    return ck


def load_checkpoint(path_or_dict, model, optimizer=None, strict=True, rank=0):
    """restore weights (+ optimizer moments, LR-schedule position and dropout RNG when an optimizer is given); returns meta."""
    ck = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, (str, bytes)) or hasattr(path_or_dict, "read") else path_or_dict
    if ck.get("meta", {}).get("format") != FORMAT:
        raise ValueError(f"unknown checkpoint format {ck.get('meta', {}).get('format')!r}")
    net = _net(model)
    sd = ck["model"]
    own = {k for k in net.state_dict().keys() if not k.startswith("clip_model.")}
    missing, unexpected = sorted(own - set(sd)), sorted(set(sd) - own)
    if strict and (missing or unexpected):
        raise KeyError(f"checkpoint/model mismatch: missing {missing[:4]}, unexpected {unexpected[:4]}")
    with torch.no_grad():
        for name, p in list(net.named_parameters()) + list(net.named_buffers()):
            if name in sd:
                if tuple(sd[name].shape) != tuple(p.shape):
                    raise ValueError(f"shape mismatch for {name}: {tuple(sd[name].shape)} vs {tuple(p.shape)}")
                p.data.copy_(sd[name])
    if getattr(net, "arena", None) is not None:
        net.arena.refresh_shadow()                       # bf16 compute copies follow the restored fp32 masters
    if optimizer is not None and "optimizer" in ck:
        a, o = optimizer.arena, ck["optimizer"]
        a.exp_avg.zero_(); a.exp_avg_sq.zero_()
        for name, p in _named_trainable(net):
            if name in o["exp_avg"] and id(p) in a.slots:
                off, n, _ = a.slots[id(p)]
                a.exp_avg[off:off + n].copy_(o["exp_avg"][name].reshape(-1))
                a.exp_avg_sq[off:off + n].copy_(o["exp_avg_sq"][name].reshape(-1))
        optimizer.hyper.copy_(torch.tensor([o["lr"], float(o["step"])]))
        r = ck.get("rng")
        if r is not None:
            seed = r["seed"] if "seed" in r else r["base"]          # "base": checkpoints written before the per-rank fix
            ops.Rng.base = (int(seed) + int(rank)) & 0xFFFFFFFF      # per-rank base: replicas must not share dropout masks
            ops.Rng.counter = r["counter"]
            if optimizer.hyper.is_cuda:
                ops.Rng.device_counter().fill_(r["device_counter"])
    return ck["meta"]
