"""Flat parameter arena: every parameter of a model lives in ONE fp32 buffer (master) with a bf16
shadow (what the GEMMs read), and — for trainable models — one fp32 gradient buffer and the two
AdamW moment buffers of the same layout.

Why (MI355X-first): 288 GB of HBM make full replication trivial, a single fused AdamW
launch streams the whole optimizer state at HBM rate, the DDP gradient all-reduce works on
contiguous slices with no flatten/unflatten copies, and q/k/v projection weights sit next to each
other so one GEMM with N = 3d serves all three.  nn.Parameter objects keep the reference's names
(state_dict-compatible with MFULL); their .data are views into the arena.
"""
import torch

from . import kernels as K

ALIGN = 32          # elements: keeps every slot 64 B (bf16) / 128 B (fp32) aligned


def _round(n, a=ALIGN):
    return (n + a - 1) // a * a


class ParamArena:
    def __init__(self, model, device, trainable=True, pad_rows=None):
        """pad_rows: {id(param): padded_row_count} — e.g. the tied embedding padded to a multiple of 32 rows so the
        LM-head dgrad can run with K = V_pad (extra rows stay exactly zero)."""
        pad_rows = pad_rows or {}
        group_of = {}
        for m in model.modules():
            if hasattr(m, "arena_groups"):
                for g in m.arena_groups():
                    for p in g:
                        group_of.setdefault(id(p), g)
        groups, seen = [], set()
        for p in model.parameters():            # registration order; a fused group is placed where its first member appears
            if id(p) in seen:
                continue
            g = [q for q in group_of.get(id(p), [p]) if id(q) not in seen]
            groups.append(g)
            seen.update(id(q) for q in g)
        self.slots = {}
        off = 0
        for g in groups:
            off = _round(off)
            for p in g:
                n = p.numel()
                if id(p) in pad_rows:
                    n = pad_rows[id(p)] * p.shape[1]
                self.slots[id(p)] = (off, p.numel(), n)
                off += n
        self.n = _round(off, 1024)
        self.device = torch.device(device)
        self.trainable = trainable
        self.flat32 = torch.zeros(self.n, device=self.device, dtype=torch.float32)
        self.flat16 = torch.zeros(self.n, device=self.device, dtype=torch.bfloat16)
        self.grad = torch.zeros(self.n, device=self.device, dtype=torch.float32) if trainable else None
        self.exp_avg = self.exp_avg_sq = None
        self.refresh_hooks = []            # callables re-deriving packed copies of weights after refresh_shadow()
        self.params = []
        for p in model.parameters():
            if id(p) in {id(q) for q in self.params}:
                continue
            o, n, _ = self.slots[id(p)]
            self.flat32[o:o + n].copy_(p.data.reshape(-1).to(self.device, torch.float32))
            p.data = self.flat32[o:o + n].view(p.shape)
            p.w16 = self.flat16[o:o + n].view(p.shape)
            if trainable and p.requires_grad:
                p.grad = self.grad[o:o + n].view(p.shape)
            self.params.append(p)
        self.refresh_shadow()
        for m in model.modules():
            if hasattr(m, "bind_arena"):
                m.bind_arena(self)

    # ---- views ---------------------------------------------------------------------------------
    def offset(self, p):
        return self.slots[id(p)][0]

    def view16(self, p, rows=None):
        """bf16 shadow of p, optionally with padded row count."""
        o, n, cap = self.slots[id(p)]
        if rows is None:
            return self.flat16[o:o + n].view(p.shape)
        assert rows * p.shape[1] <= cap
        return self.flat16[o:o + rows * p.shape[1]].view(rows, p.shape[1])

    def fused(self, plist, which):
        """contiguous view over adjacent parameters (e.g. [k,v,q] weights -> [3d, d])."""
        o0 = self.offset(plist[0])
        tot, o = 0, o0
        for p in plist:
            assert self.offset(p) == o, "parameters are not adjacent in the arena"
            o += p.numel(); tot += p.numel()
        buf = {"w16": self.flat16, "f32": self.flat32, "grad": self.grad}[which]
        if buf is None:
            return None
        v = buf[o0:o0 + tot]
        if plist[0].dim() == 2:
            return v.view(-1, plist[0].shape[1])
        return v

    # ---- maintenance ---------------------------------------------------------------------------
    def refresh_shadow(self):
        if self.device.type == "cuda":
            K.cast_f32_bf16(self.flat32, self.flat16)
        else:
            # host-side arenas exist only for layout / reducer tests (gloo); no compute op accepts CPU tensors
            self.flat16.copy_(self.flat32)
        for hook in self.refresh_hooks:
            hook()

    def init_optimizer_state(self):
        self.exp_avg = torch.zeros(self.n, device=self.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.n, device=self.device, dtype=torch.float32)

    def bucket_slices(self, bucket_bytes=256 << 20):
        """contiguous [start, end) element ranges of the gradient arena, in REVERSE layout order
        (the order backward finishes them), for the DDP reducer."""
        per = max(4, bucket_bytes // 16 * 4)          # whole 16-byte groups: a bucket is also a range of the vectorised AdamW kernel
        out, end = [], self.n
        while end > 0:
            start = max(0, end - per)
            out.append((start, end))
            end = start
        return out
