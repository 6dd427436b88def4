"""Caption generation on gfx950 kernels: encoder once, then a KV-cached single-token decoder loop with beam
search — the `model.generate(input_ids, attention_mask, num_beams=5, max_length=50, image_features=..., face_features=...,
name_ids=..., add_ner_ffn=True[, length_penalty=2.0])` call of TRAIN:513-520 / DDPINF:758-842.

What runs where
  * device (hand-written kernels): embed+pos+LN of the new token, fused k|v|q projection, KV-cache append, fused
    attention over the cache (self) and over the per-layer cross K/V computed once from the encoder states, FFN,
    LM head, log-softmax + logits processors + top-2k (`beam_topk`), cache reorder for all layers in one launch
    (`gather_rows`; _reorder_cache MFULL:2066-2074 — cross K/V are never reordered, as in the reference).
  * beam bookkeeping (transformers 4.18 BeamSearchScorer.process: n-best lists, EOS handling, done flags) and the
    NoRepeatNGram ban lists: `vacnic_beam_step`, one wave per batch item, in the same per-position graph; the
    host reads the state ONCE after the last position (and polls the done flags every 8th position to stop early).
    `device_beams=False` keeps the host-side scorer (one device->host copy of the [B*beams, 2*beams] candidates per token).

The cache is preallocated ([layers, B*beams, max_length, 2d] bf16, k|v interleaved per row) instead of the reference's
per-step torch.cat (MFULL:489-492).
"""
import os

import torch

from . import _lib, kernels as K

BF16 = torch.bfloat16


class GraphedCall:
    """fn(*tensors) -> tensor(s), captured as a hipGraph per input signature (shapes + dtypes) and replayed: at batch 1 the
    encoder / ViT of a caption are ~600 launches of a few microseconds of work each, i.e. bound by the Python launch path
    (5.5 + 2.5 ms per caption eagerly).  First sighting of a signature runs eagerly (warms kernels and lazy buffers), the second
    captures, later ones copy the inputs into the graph's static buffers and replay.  The outputs live in the graph's pool and
    are overwritten by the next replay — callers consume (or clone) them before calling again.  A body that stream capture
    rejects (host synchronisation inside) stays eager for that signature, with a warning; any other error is raised."""

    def __init__(self, fn, max_entries=8):
        self.fn, self.cache, self.max_entries = fn, {}, max_entries

    def __call__(self, *args):
        from . import streams
        # the captured launches depend on the stream schedule in force; weights are read in place from the arenas (same
        # addresses after an optimizer step or a reload), so they are not part of the key
        key = (streams.enabled(),) + tuple((tuple(a.shape), a.dtype) if a is not None else None for a in args)
        ent = self.cache.get(key)
        if ent is None:
            if len(self.cache) >= self.max_entries:
                self.cache.pop(next(iter(self.cache)))
            self.cache[key] = "seen"
            return self.fn(*args)
        if ent == "eager":
            return self.fn(*args)
        if ent == "seen":
            static = [a.clone() if a is not None else None for a in args]
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            try:
                # (this graph may replay beside other graphs — CaptionPipeline — so its split-K fix-up buffers are its own)
                with K.capture_scope(("graphed-call", id(g))), torch.cuda.graph(g):
                    out = self.fn(*static)
            except _lib.VacnicError:
                raise                                     # a kernel / argument error is a bug, not a capture limitation
            except RuntimeError as e:
                # only what stream capture itself rejects (a host synchronisation inside the body, an unsupported call while
                # capturing) makes the signature eager; anything else is passed on
                torch.cuda.synchronize()
                msg = str(e).lower()
                if not any(w in msg for w in ("captur", "hipgraph", "cudagraph", "graph")):
                    raise
                import warnings
                warnings.warn(f"GraphedCall: {getattr(self.fn, '__name__', 'fn')} stays eager for signature {key}: {e}")
                self.cache[key] = "eager"
                return self.fn(*args)
            ent = self.cache[key] = (g, static, out)
        g, static, out = ent
        for st, a in zip(static, args):
            if st is not None:
                st.copy_(a, non_blocking=True)
        g.replay()
        return out


class _BeamHyps:
    """n-best list of finished hypotheses, scored sum_logprobs / len**length_penalty (len counts the decoder start token,
    not the closing EOS) — transformers==4.18 BeamHypotheses."""

    def __init__(self, n, length_penalty, early_stopping):
        self.n, self.lp, self.early = n, length_penalty, early_stopping
        self.beams, self.worst = [], 1e9

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (len(hyp) ** self.lp)
        if len(self.beams) < self.n or score > self.worst:
            self.beams.append((score, list(hyp)))
            if len(self.beams) > self.n:
                srt = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst = srt[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.beams) < self.n:
            return False
        if self.early:
            return True
        return self.worst >= best_sum_logprobs / cur_len ** self.lp


def _banned(seq, n):
    if n <= 0 or len(seq) + 1 < n:
        return []
    prefix = tuple(seq[len(seq) - (n - 1):]) if n > 1 else ()
    return [seq[i + n - 1] for i in range(len(seq) - n + 1) if tuple(seq[i:i + n - 1]) == prefix]


class CachedDecoder:
    """Single-token decoder over preallocated KV caches (eval mode, no autograd).  All buffers are allocated once and the
    cache in use at position t is `cache[t & 1]` when beams are reordered every step (no hidden toggle), so a step is a pure
    function of (static buffers, t) and can be captured as a hipGraph."""

    def __init__(self, model, rows, S, max_length, reorders):
        self.m = model
        dec = model.model.decoder
        self.dec = dec
        self.L = len(dec.layers)
        d = model.config.d_model
        self.d, self.H = d, model.config.decoder_attention_heads
        self.rows, self.Tmax, self.S, self.reorders = rows, max_length, S, reorders
        dev = model.emb16_pad.device
        # one extra position per row: the fused k|v|q projection of position t is written IN PLACE at [row, t, 0:3d] — k|v
        # land in their cache slot, q parks in the first d columns of position t+1 (overwritten by that position's own k
        # before anything reads it), so no append copy is needed (MFULL:489-492 does torch.cat per step)
        self.cache = [torch.zeros((self.L, rows, max_length + 1, 2 * d), device=dev, dtype=BF16) for _ in range(2 if reorders else 1)]
        self.enc_b = torch.empty((rows, S, d), device=dev, dtype=BF16)
        self.enc_mask = torch.empty((rows, S), device=dev, dtype=torch.uint8)
        self.cross_all = torch.empty((self.L, rows, S, 2 * d), device=dev, dtype=BF16)      # one block: a staged caption arrives in one copy
        self.cross = [self.cross_all[li] for li in range(self.L)]
        self.lidx = (torch.arange(self.L, device=dev)[:, None] * rows).contiguous()
        # persistent decoder-step kernel (vacnic_decoder_step): all layers of a position in one launch.  VACNIC_DECODE_PER_OP=1
        # keeps the kernel-per-op chain (the reference the step kernel is tested against).
        F = dec.layers[0].fc1.weight.shape[0]
        self.F = F
        self.step_kernel = (os.environ.get("VACNIC_DECODE_PER_OP", "0") != "1" and rows <= 8 and d <= 1024 and d % 8 == 0 and F <= 4096
                            and F % 8 == 0 and self.H * 64 == d and max_length <= 8191)
        self.step_table = None
        self.trace, self.trace_wg = None, 0       # tools/decstep_trace.py: per-phase time stamps of one workgroup
        self.last_hidden = None
        if self.step_kernel:
            self.sync = torch.zeros(int(_lib.lib.vacnic_decoder_step_sync_bytes()) // 4, device=dev, dtype=torch.int32)
            # tagged-slot exchange (no grid barriers) when the workgroup count fits the GPU; VACNIC_DECODE_BARRIER=1: barrier variant
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            G = max(d // 4, F // 16)
            self.slots = None
            if (os.environ.get("VACNIC_DECODE_BARRIER", "0") != "1" and d % 16 == 0 and F % 16 == 0 and G <= min(256, cus)
                    and rows * self.H <= min(128, G) and self.L <= 120):
                self.slots = torch.zeros(int(_lib.lib.vacnic_decoder_step_slots_bytes(self.L)), device=dev, dtype=torch.uint8)
            self.hbuf = [torch.zeros((rows, d), device=dev, dtype=BF16) for _ in range(2)]
            self.obuf, self.ctxb, self.qbuf = (torch.zeros((rows, d), device=dev, dtype=BF16) for _ in range(3))
            self.fbuf = torch.zeros((rows, F), device=dev, dtype=BF16)

    def begin(self, enc_h, mask_u8, nb):
        """expand the encoder states to beams (HF _expand_inputs_for_generation: row b*nb + j <- batch b) and compute the
        cross-attention K/V of every layer once (the decoder's largest GEMMs: M = rows*S)."""
        B, S, d = enc_h.shape
        for j in range(nb):
            K.copy3d(enc_h, self.enc_b[j::nb], B, S, d)
        self.enc_mask.copy_(mask_u8.repeat_interleave(nb, dim=0))
        # one caption (B == 1): all beams attend to the SAME source, so the cross-attention K/V are projected once for the single
        # row and every beam reads them through a batch stride of 0 (5x fewer bytes per token, L2-resident after the first beam)
        self.shared_kv = B == 1
        src, M = (enc_h, S) if self.shared_kv else (self.enc_b, self.rows * S)
        for li, layer in enumerate(self.dec.layers):
            a = layer.encoder_attn
            K.gemm(src.reshape(M, d), a.s_kv.w16, M, 2 * d, d, bias=a.s_kv.bias, out=self.cross[li].view(self.rows * S, 2 * d)[:M])

    def begin_staged(self, stage):
        """begin() for a caption whose encoder side ran ahead of time (EncodeStage, B == 1): the cross-attention K/V of every layer
        and the key mask are copied from the staging buffers (24 MB device to device at BART-large: ~15 us)."""
        if self.rows != stage.nb or self.S != stage.S:
            raise ValueError("begin_staged: the stage was encoded for another decode shape")
        self.shared_kv = True
        S, d = self.S, self.d
        K.copy3d(stage.cross_all, self.cross_all[:, 0], self.L, S, 2 * d)
        self.enc_mask.copy_(stage.mask.expand(self.rows, S))

    def cache_at(self, t):
        return self.cache[t & 1] if self.reorders else self.cache[0]

    def _layer_table(self):
        """device array of vacnic_decoder_layer (weights / biases / LayerNorm parameters / cross K|V of every layer)."""
        if self.step_table is None or self.step_table[1] != self.shared_kv:
            d, S = self.d, self.S
            arr = (_lib.DecoderLayer * self.L)()
            for li, layer in enumerate(self.dec.layers):
                a, c = layer.self_attn, layer.encoder_attn
                mats = {"w_kvq": (a.s_kvq.w16, 3 * d, d), "w_so": (a.s_out.w16, d, d), "w_cq": (c.s_q.w16, d, d), "w_co": (c.s_out.w16, d, d),
                        "w_fc1": (layer.s_fc1.w16, self.F, d), "w_fc2": (layer.s_fc2.w16, d, self.F)}
                for name, (w, n, k) in mats.items():
                    if tuple(w.shape) != (n, k) or not w.is_contiguous():
                        raise ValueError(f"decoder_step: {name} of layer {li} must be a contiguous [{n}, {k}] matrix, got {tuple(w.shape)}")
                    setattr(arr[li], name, w.data_ptr())
                for name, b in (("b_kvq", a.s_kvq.bias), ("b_so", a.s_out.bias), ("b_cq", c.s_q.bias), ("b_co", c.s_out.bias),
                                ("b_fc1", layer.s_fc1.bias), ("b_fc2", layer.s_fc2.bias)):
                    setattr(arr[li], name, b.data_ptr() if b is not None else None)
                for name, ln in (("ln_self", layer.self_attn_layer_norm), ("ln_cross", layer.encoder_attn_layer_norm),
                                 ("ln_final", layer.final_layer_norm)):
                    setattr(arr[li], name + "_g", ln.weight.data.data_ptr())
                    setattr(arr[li], name + "_b", ln.bias.data.data_ptr())
                arr[li].cross_kv = self.cross[li].data_ptr()
                arr[li].cross_bs = 0 if self.shared_kv else S * 2 * d
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.step_table = (host.to(self.cache[0].device), self.shared_kv)
        return self.step_table[0]

    def _step_all_layers(self, h, t):
        """every decoder layer of position t in one launch; returns the last layer's (block output, residual) pair."""
        lay = self.dec.layers[0]
        _lib.call_struct("vacnic_decoder_step", stream=K._stream(), layers=self._layer_table().data_ptr(), cache=self.cache_at(t).data_ptr(),
                         h0=h.data_ptr(), hbuf0=self.hbuf[0].data_ptr(), hbuf1=self.hbuf[1].data_ptr(), obuf=self.obuf.data_ptr(),
                         ctx=self.ctxb.data_ptr(), qbuf=self.qbuf.data_ptr(), fbuf=self.fbuf.data_ptr(), enc_mask=self.enc_mask.data_ptr(),
                         sync=self.sync.data_ptr(), slots=self.slots.data_ptr() if self.slots is not None else None, L=self.L, R=self.rows, d=self.d, H=self.H, F=self.F, S=self.S, t=t, Tmax=self.Tmax,
                         eps=lay.final_layer_norm.eps, scale=0.125, trace=self.trace.data_ptr() if self.trace is not None else None,
                         trace_wg=self.trace_wg)
        return self.obuf, self.hbuf[self.L & 1]

    def check_step_kernel(self):
        """after the results were read back: did a grid barrier of the step kernel time out?"""
        if self.step_kernel and int(self.sync[-64].item()) != 0:
            # a timed-out launch left the barrier counters / nonce and (slot variant) units tagged with the aborted nonce behind:
            # clear BOTH, so that a later launch can never match stale tags, and take this decoder off the step kernel — whoever
            # keeps using it (a cached DecodeSession) continues on the kernel-per-op chain
            self.sync.zero_()
            if self.slots is not None:
                self.slots.zero_()
            self.step_kernel = False
            self.step_table = None
            raise RuntimeError("vacnic_decoder_step: a wait inside the step kernel timed out (workgroups not co-resident?); results of "
                               "this caption are invalid; this decoder falls back to the kernel-per-op chain")

    def step(self, ids_t, t, hidden_only=False):
        """ids_t int64 [rows, 1] (token at position t) -> fp32 logits [rows, V_pad]; hidden_only: the decoder's final hidden rows
        [rows, d] (bf16, after the last layer's LayerNorm) instead — the input of kernels.lmhead_topk."""
        m, dec, d, H, R = self.m, self.dec, self.d, self.H, self.rows
        ln = dec.layernorm_embedding
        h, _, _ = K.embed_ln_fwd(ids_t, dec.embed_tokens.weight.w16, dec.embed_positions.weight.w16, ln.weight.data, ln.bias.data,
                                 embed_scale=dec.embed_scale, pos_offset=2 + t)
        cache = self.cache_at(t)
        # R <= 8 rows (one caption, <= 8 beams): every post-LN (residual add + LayerNorm) rides as a prologue on the projection
        # that consumes it (vacnic_gemv_ln_bf16) — 36 fewer launches per token, bit-identical values.  `pend` = the block output
        # and residual whose LayerNorm has not been applied yet.
        fuse = R <= 8 and d <= 1024 and d % 8 == 0
        pend = None

        def ln_then(lnm, w, N, bias, out=None, ldo=None, act=None, out_mode=0):
            o_, res_ = pend
            hn = torch.empty((R, d), device=o_.device, dtype=BF16)
            y = K.gemv_ln(o_.view(R, d), res_.view(R, d), lnm.weight.data, lnm.bias.data, w, R, N, d, bias=bias, out=out, ln_out=hn,
                          ldw=w.stride(0), ldo=ldo, act=act, out_mode=out_mode, eps=lnm.eps)
            return y, hn

        if self.step_kernel:
            pend = self._step_all_layers(h.view(R, d), t)
            pend_ln = dec.layers[-1].final_layer_norm
        for li, layer in enumerate(dec.layers if not self.step_kernel else ()):
            a = layer.self_attn
            row = cache[li][:, t]                                                               # [R, 2d] view, row stride (Tmax+1)*2d
            if pend is None:
                K.gemm(h.view(R, d), a.s_kvq.w16, R, 3 * d, d, bias=a.s_kvq.bias, out=row, ldo=row.stride(0))   # k|v|q in place
            else:
                _, h = ln_then(pend_ln, a.s_kvq.w16, 3 * d, a.s_kvq.bias, out=row, ldo=row.stride(0))
            q = cache[li][:, t + 1:t + 2, :d]
            ctx, _ = K.attn_fwd(q, cache[li][:, :t + 1, :d], cache[li][:, :t + 1, d:], R, H, 1, t + 1, need_lse=False)
            o = K.gemm(ctx.view(R, d), a.s_out.w16, R, d, d, bias=a.s_out.bias)
            c = layer.encoder_attn
            if fuse:
                pend = (o, h)
                q, h = ln_then(layer.self_attn_layer_norm, c.s_q.w16, d, c.s_q.bias)
                q = q.view(R, 1, d)
            else:
                h, _, _ = K.add_ln_fwd(o.view(R, 1, d), h, layer.self_attn_layer_norm.weight.data, layer.self_attn_layer_norm.bias.data,
                                       need_stats=False)
                q = K.gemm(h.view(R, d), c.s_q.w16, R, d, d, bias=c.s_q.bias).view(R, 1, d)
            kv = self.cross[li][:1].expand(R, -1, -1) if self.shared_kv else self.cross[li]
            ctx, _ = K.attn_fwd(q, kv[..., :d], kv[..., d:], R, H, 1, kv.shape[1], key_mask=self.enc_mask, need_lse=False)
            o = K.gemm(ctx.view(R, d), c.s_out.w16, R, d, d, bias=c.s_out.bias)
            if fuse:
                pend = (o, h)
                f, h = ln_then(layer.encoder_attn_layer_norm, layer.s_fc1.w16, layer.s_fc1.N, layer.s_fc1.bias, act="gelu")
            else:
                h, _, _ = K.add_ln_fwd(o.view(R, 1, d), h, layer.encoder_attn_layer_norm.weight.data, layer.encoder_attn_layer_norm.bias.data,
                                       need_stats=False)
                f = K.gemm(h.view(R, d), layer.s_fc1.w16, R, layer.s_fc1.N, d, bias=layer.s_fc1.bias, act="gelu")
            o = K.gemm(f, layer.s_fc2.w16, R, d, layer.s_fc2.K, bias=layer.s_fc2.bias)
            if fuse:
                pend, pend_ln = (o, h), layer.final_layer_norm
            else:
                h, _, _ = K.add_ln_fwd(o.view(R, 1, d), h, layer.final_layer_norm.weight.data, layer.final_layer_norm.bias.data, need_stats=False)
        if hidden_only:
            if self.step_kernel and self.slots is not None:
                self.last_hidden = pend[0]
            elif fuse and pend is not None:
                o_, res_ = pend
                hn, _, _ = K.add_ln_fwd(o_.view(R, 1, d), res_.view(R, 1, d), pend_ln.weight.data, pend_ln.bias.data, need_stats=False)
                self.last_hidden = hn.view(R, d)
            else:
                self.last_hidden = h.view(R, d)
            return self.last_hidden
        logits = torch.empty((R, m.V_pad), device=h.device, dtype=torch.float32)
        # last_hidden: the final hidden rows [R, d] behind these logits (diagnostics / fixture construction; overwritten every step)
        if self.step_kernel and self.slots is not None:     # slot variant: obuf already holds the final (normalised) hidden rows
            K.gemm(pend[0], m.emb16_pad, R, m.V, d, bias=m.final_logits_bias.view(-1), out=logits, ldo=m.V_pad, out_mode=1)
            self.last_hidden = pend[0]
        elif fuse and pend is not None:
            _, self.last_hidden = ln_then(pend_ln, m.emb16_pad, m.V, m.final_logits_bias.view(-1), out=logits, ldo=m.V_pad, out_mode=1)
        else:
            K.gemm(h.view(R, d), m.emb16_pad, R, m.V, d, bias=m.final_logits_bias.view(-1), out=logits, ldo=m.V_pad, out_mode=1)
            self.last_hidden = h.view(R, d)
        return logits

    def reorder(self, beam_idx, t):
        """before step t: the self-attention caches of all layers follow their beams (one launch; _reorder_cache
        MFULL:2066-2074): cache[(t-1)&1] gathered into cache[t&1]."""
        L, R = self.L, self.rows
        # one permutation for every layer's block of R rows (period = R), and only the t positions filled so far (+ the slot where
        # the per-op path parks q) travel: half the bytes of whole rows on average, no index arithmetic on the host side
        K.gather_rows(self.cache[(t - 1) & 1], self.cache[t & 1], beam_idx, L * R, (t + 1) * 2 * self.d * 2,
                      row_stride_bytes=(self.Tmax + 1) * 2 * self.d * 2, period=R)


class DecodeSession:
    """Everything device-side of one decode shape (rows, S, max_length, logits-processor options): static buffers, the
    cached decoder, and one hipGraph per position t = {cache reorder, 12-layer single-token decoder, LM head, log-softmax +
    processors + top-2k}.  The first caption of a shape runs eagerly (it also warms every kernel); from the second on each
    position is captured on first use and replayed afterwards, which removes the ~170 Python->HIP launches per token that
    bound the eager loop."""

    def __init__(self, model, R, S, max_length, nb, ngram, min_length, forced_eos, eos, forced_bos=None):
        self.model, self.R, self.nb, self.max_length = model, R, nb, max_length
        self.ngram, self.min_length, self.forced_eos, self.eos = ngram, min_length, forced_eos, eos
        self.forced_bos = forced_bos
        dev = model.emb16_pad.device
        self.dec = CachedDecoder(model, R, S, max_length, reorders=nb > 1)
        self.ids_s = torch.zeros((R, 1), device=dev, dtype=torch.long)
        self.scores_s = torch.zeros(R, device=dev, dtype=torch.float32)
        self.src_s = torch.zeros(R, device=dev, dtype=torch.long)
        self.bans_s = torch.full((R, max_length), -1, device=dev, dtype=torch.int32) if ngram > 0 else None
        self.h_ids = torch.zeros((R, 1), dtype=torch.long).pin_memory()
        self.h_scores = torch.zeros(R, dtype=torch.float32).pin_memory()
        self.h_src = torch.zeros(R, dtype=torch.long).pin_memory()
        self.h_bans = torch.full((R, max_length), -1, dtype=torch.int32).pin_memory() if ngram > 0 else None
        self.graphs, self.outs, self.pool = {}, {}, None
        # few rows (one caption's beams): the fused LM head + top-k (VACNIC_FUSED_LMHEAD_TOPK=0: skinny GEMM -> vacnic_beam_topk)
        self.fused_tail = (R <= 8 and model.emb16_pad.shape[1] <= 1024 and model.emb16_pad.shape[1] % 8 == 0 and min(2 * nb, model.V) <= 64
                           and model.V <= 512 * 128 and os.environ.get("VACNIC_FUSED_LMHEAD_TOPK", "1") != "0")
        self.captions = 0
        self.beam = None                  # device-side beam bookkeeping (enable_device_beams)

    def enable_device_beams(self, B, length_penalty, early_stopping, pad):
        """state of vacnic_beam_step: token histories, n-best lists and done flags live in HBM; the step kernel
        also writes this session's next-token / beam-score / reorder-index / ban buffers, so a position needs no host round trip."""
        from . import _lib
        dev = self.ids_s.device
        R, nb, L = self.R, self.nb, self.max_length
        i32 = dict(device=dev, dtype=torch.int32)
        if self.bans_s is None and self.ngram > 0:
            self.bans_s = torch.full((R, L), -1, **i32)
        self.seq = torch.zeros((2, R, L), **i32)
        self.done_d = torch.zeros(B, **i32)
        self.h_done = torch.zeros(B, dtype=torch.int32).pin_memory()
        self.hyp_cnt = torch.zeros(B, **i32)
        self.hyp_worst = torch.zeros(B, device=dev, dtype=torch.float64)
        self.hyp_score = torch.zeros((B, nb), device=dev, dtype=torch.float64)
        self.hyp_len = torch.zeros((B, nb), **i32)
        self.hyp_seq = torch.zeros((B, nb, L), **i32)
        self.B, self.lp, self.early, self.pad = B, length_penalty, early_stopping, pad
        self.beam = _lib.BeamState(seq0=self.seq[0].data_ptr(), seq1=self.seq[1].data_ptr(), beam_scores=self.scores_s.data_ptr(),
                                   done=self.done_d.data_ptr(), hyp_cnt=self.hyp_cnt.data_ptr(), hyp_worst=self.hyp_worst.data_ptr(),
                                   hyp_score=self.hyp_score.data_ptr(), hyp_len=self.hyp_len.data_ptr(), hyp_seq=self.hyp_seq.data_ptr(),
                                   next_ids=self.ids_s.data_ptr(), src_idx=self.src_s.data_ptr(),
                                   bans=self.bans_s.data_ptr() if self.bans_s is not None else None,
                                   B=B, nb=nb, Lmax=L, V=self.model.V, eos=self.eos, pad=pad, no_repeat_ngram_size=self.ngram,
                                   early_stopping=int(bool(early_stopping)), length_penalty=float(length_penalty))

    def run_device(self, start, use_graphs):
        """the whole beam search of one batch without per-token host synchronisation; returns the host-side tensors finalize needs."""
        from . import _lib
        st = K._stream()
        _lib.check(_lib.lib.vacnic_beam_init(_lib.C.byref(self.beam), int(start), st))
        steps = 0
        pending = None                                       # (event, pinned copy of the done flags): polled without blocking the launches
        for t in range(self.max_length - 1):
            if use_graphs and self.captions > 0:
                g = self.graphs.get(t)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    with K.capture_scope(("decode-session", id(self))), torch.cuda.graph(g, pool=self.pool):
                        self.body(t)
                    if self.pool is None:
                        self.pool = g.pool()
                    self.graphs[t] = g
                g.replay()
            else:
                self.body(t)
            steps = t + 1
            # early stop: the done flags are copied out every 8th position and looked at when the copy has landed, so the launch
            # path never waits; positions enqueued in the meantime only pad finished items (BeamSearchScorer.process on done batches)
            if pending is not None and pending[0].query():
                if bool(pending[1].all()):
                    break
                pending = None
            if pending is None and t % 8 == 7 and t + 1 < self.max_length - 1:
                self.h_done.copy_(self.done_d, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending = (ev, self.h_done)
        cur_len = steps + 1                                  # tokens in every live history
        seqs = self.seq[steps & 1].cpu()
        self.dec.check_step_kernel()
        return (cur_len, seqs, self.scores_s.cpu(), self.done_d.cpu(), self.hyp_cnt.cpu(), self.hyp_score.cpu(), self.hyp_len.cpu(),
                self.hyp_seq.cpu())

    def body(self, t):
        if self.nb > 1 and t > 0:
            self.dec.reorder(self.src_s, t)
        cur_len = t + 1
        forced = self.forced_eos if (self.forced_eos is not None and cur_len == self.max_length - 1) else -1
        if self.forced_bos is not None and cur_len == 1:             # ForcedBOSTokenLogitsProcessor (runs before ForcedEOS in HF)
            forced = self.forced_bos if forced < 0 else forced
        V = self.model.V
        if self.fused_tail:
            # LM head + log_softmax + processors + top-2nb in one C call, no [R, V] logits (a forced position needs no LM head at all)
            hid = self.dec.step(self.ids_s, t, hidden_only=True)
            tv, ti = K.lmhead_topk(hid, self.model.emb16_pad, V, min(2 * self.nb, V), bias=self.model.final_logits_bias.view(-1),
                                   beam_scores=self.scores_s, bans=self.bans_s, eos=self.eos, suppress_eos=cur_len < self.min_length,
                                   forced_token=forced)
        else:
            logits = self.dec.step(self.ids_s, t)
            tv, ti = K.beam_topk(logits, V, min(2 * self.nb, V), beam_scores=self.scores_s, bans=self.bans_s, eos=self.eos,
                                 suppress_eos=cur_len < self.min_length, forced_token=forced)
        if self.beam is not None:
            from . import _lib
            _lib.check(_lib.lib.vacnic_beam_step(_lib.C.byref(self.beam), tv.data_ptr(), ti.data_ptr(), tv.shape[1], cur_len, K._stream()))
        return tv, ti

    def step(self, t, last_tokens, scores, src, bans, use_graphs):
        """host lists in, (top values, top ids) on the host out — the one device->host sync of the position."""
        self.h_ids[:, 0] = torch.as_tensor(last_tokens)
        self.ids_s.copy_(self.h_ids, non_blocking=True)
        self.h_scores.copy_(torch.as_tensor(scores, dtype=torch.float32))
        self.scores_s.copy_(self.h_scores, non_blocking=True)
        if self.nb > 1 and t > 0:
            self.h_src.copy_(torch.as_tensor(src))
            self.src_s.copy_(self.h_src, non_blocking=True)
        if self.bans_s is not None:
            self.h_bans.fill_(-1)
            for r, bl in enumerate(bans):
                if bl:
                    self.h_bans[r, :len(bl)] = torch.as_tensor(bl, dtype=torch.int32)
            self.bans_s.copy_(self.h_bans, non_blocking=True)
        if use_graphs and self.captions > 0:
            g = self.graphs.get(t)
            if g is None:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with K.capture_scope(("decode-session", id(self))), torch.cuda.graph(g, pool=self.pool):
                    self.outs[t] = self.body(t)
                if self.pool is None:
                    self.pool = g.pool()
                self.graphs[t] = g
            g.replay()
            tv, ti = self.outs[t]
        else:
            tv, ti = self.body(t)
        tv, ti = tv.cpu(), ti.cpu()
        self.dec.check_step_kernel()
        return tv, ti


def _run_encoder(model, use_graphs, add_ner_ffn, input_ids, mask_u8, image_features, name_ids, name_mask, face_features, face_mask):
    """the encoder pass of one generate() call: a hipGraph per input signature (GraphedCall) unless use_graphs is off"""
    def run_encoder(ids, m8, img, nids, nmask, faces, fmask, add_ner_ffn=add_ner_ffn):
        return model.model.encoder(input_ids=ids, attention_mask=m8, image_features=img, name_ids=nids, name_mask=nmask,
                                   face_features=faces, face_mask=fmask, add_ner_ffn=add_ner_ffn)["last_hidden_state"]
    if not use_graphs:
        return run_encoder(input_ids, mask_u8, image_features, name_ids, name_mask, face_features, face_mask)
    ge = model.__dict__.setdefault("_graphed_encoders", {})
    if add_ner_ffn not in ge:
        ge[add_ner_ffn] = GraphedCall(run_encoder)
    return ge[add_ner_ffn](input_ids, mask_u8, image_features, name_ids, name_mask, face_features, face_mask)


class EncodeStage:
    """Everything the beam search of ONE caption (B == 1) needs from the encoder side, in buffers of its own: the cross-attention K/V
    of every decoder layer and the source key mask.  Filled by encode() on whatever stream is current — CaptionPipeline runs it on
    a side stream for caption i + 1 while caption i is being decoded — and consumed by generate(encoded=stage)."""

    def __init__(self, model, S, nb):
        dec = model.model.decoder
        d = model.config.d_model
        dev = model.emb16_pad.device
        self.S, self.d, self.nb = S, d, nb
        self.cross_all = torch.empty((len(dec.layers), S, 2 * d), device=dev, dtype=BF16)
        self.cross = [self.cross_all[li] for li in range(len(dec.layers))]
        self.mask = torch.empty((1, S), device=dev, dtype=torch.uint8)

    @torch.no_grad()
    def encode(self, model, input_ids, attention_mask, image_features=None, face_features=None, face_mask=None, name_ids=None, name_mask=None,
               add_ner_ffn=True, use_graphs=True):
        if input_ids.shape[0] != 1 or input_ids.shape[1] != self.S:
            raise ValueError("EncodeStage.encode: one caption of the staged source length")
        mask_u8 = attention_mask.to(torch.uint8) if attention_mask.dtype != torch.uint8 else attention_mask
        enc_h = _run_encoder(model, use_graphs, add_ner_ffn, input_ids, mask_u8, image_features, name_ids, name_mask, face_features, face_mask)
        S, d = self.S, self.d
        for li, layer in enumerate(model.model.decoder.layers):
            a = layer.encoder_attn
            K.gemm(enc_h.reshape(S, d), a.s_kv.w16, S, 2 * d, d, bias=a.s_kv.bias, out=self.cross[li])
        self.mask.copy_(mask_u8)
        return self


class CaptionPipeline:
    """model.generate over a stream of single captions (test_batch_size 1, the reference's generation loop TRAIN:480-530) as a
    pipeline of three stages on three streams: while caption i is in its beam search (a chain of latency-bound launches that
    leaves most of the GPU idle), the encoder and the cross-attention K/V projection of caption i + 1 run on a second stream and the
    masks + image tower of caption i + 2 on a third — none of them depends on the beam search.  Same ids as the sequential loop
    (the kernels are the same, only their streams differ).  `inputs_fn(batch)` -> (input_ids, attention_mask, image_features, kw)
    builds the masks and runs the image tower (training._model_inputs).  Batches with more than one caption decode through the
    plain generate() with their inputs_fn ahead."""

    def __init__(self, model, inputs_fn, num_beams, **gen_kw):
        self.model, self.inputs_fn, self.nb, self.gen_kw = model, inputs_fn, num_beams, gen_kw
        # (stream priorities change nothing: 34.7-34.8 ms per caption with one side stream either way)
        self.s_img, self.s_enc = torch.cuda.Stream(), torch.cuda.Stream()
        self.stages = {}

    def _inputs(self, batch, after):
        """stage A (image stream): masks + image tower of `batch`, not before event `after` (its tensors are on the device)"""
        with torch.cuda.stream(self.s_img):
            self.s_img.wait_event(after)
            src, src_mask, feats, kw = self.inputs_fn(batch)
            if feats is not None:
                feats = feats.clone()                       # a graphed image tower overwrites its output at its next call (which runs ahead)
            ev = torch.cuda.Event()
            ev.record(self.s_img)
        return {"batch": batch, "inputs": (src, src_mask, feats, kw), "a": ev, "stage": None, "b": ev}

    def _encode(self, ent, after):
        """stage B (encoder stream): encoder + cross K/V of one caption into the EncodeStage, not before `after` (the previous caption's
        staged data have been copied into the decoder's buffers)"""
        src, src_mask, feats, kw = ent["inputs"]
        if src.shape[0] != 1:
            return
        with torch.cuda.stream(self.s_enc):
            self.s_enc.wait_event(ent["a"])
            if after is not None:
                self.s_enc.wait_event(after)
            S = src.shape[1]
            stage = self.stages.get(S)
            if stage is None:
                stage = self.stages[S] = EncodeStage(self.model, S, self.nb)
            stage.encode(self.model, src, src_mask, image_features=feats, add_ner_ffn=self.gen_kw.get("add_ner_ffn", True), **kw)
            ev = torch.cuda.Event()
            ev.record(self.s_enc)
        ent["stage"], ent["b"] = stage, ev

    def __call__(self, batches):
        """yields (batch, generated ids) in order"""
        main = torch.cuda.current_stream()
        it = iter(batches)

        def fetch():
            b = next(it, None)
            if b is None:
                return None
            ready = torch.cuda.Event()
            ready.record(main)                              # the caller's host-to-device copies of this batch were enqueued on `main`
            return self._inputs(b, ready)

        cur = fetch()
        if cur is None:
            return
        self._encode(cur, None)
        nxt = fetch()
        while cur is not None:
            main.wait_event(cur["b"])
            src, src_mask, feats, kw = cur["inputs"]
            ahead = {}

            def advance(copied):
                # the next caption's encoder side may overwrite the stage once this caption's copy out of it is enqueued (`copied`);
                # the caption after that gets its image tower
                if nxt is not None:
                    self._encode(nxt, copied)
                    ahead["e"] = fetch()

            if cur["stage"] is not None:
                def hook():
                    ev = torch.cuda.Event()
                    ev.record(main)
                    advance(ev)
                gen = self.model.generate(input_ids=src, attention_mask=src_mask, num_beams=self.nb, encoded=cur["stage"], _after_begin=hook,
                                          **self.gen_kw)
            else:
                advance(None)
                gen = self.model.generate(input_ids=src, attention_mask=src_mask, num_beams=self.nb, image_features=feats, **kw, **self.gen_kw)
            yield cur["batch"], gen
            cur, nxt = nxt, ahead.get("e")


@torch.no_grad()
def generate(model, input_ids=None, attention_mask=None, num_beams=1, max_length=20, length_penalty=1.0, early_stopping=False,
             no_repeat_ngram_size=0, min_length=0, forced_eos_token_id="config", forced_bos_token_id=None, image_features=None,
             face_features=None, face_mask=None, name_ids=None, name_mask=None, add_ner_ffn=True, use_graphs=True, device_beams=True,
             return_nbest=False, encoded=None, _after_begin=None, **unused):
    """GenerationMixin.generate(do_sample=False) semantics of transformers 4.18 for this model (greedy = 1 beam).
    Returns int64 [B, L] starting with decoder_start_token_id, padded with pad_token_id.
    Keyword defaults are the LIBRARY defaults; the defaults a hub checkpoint's config.json adds on top (what the reference's
    `model.generate(num_beams, max_length)` inherits after from_pretrained, TRAIN:513-520) are config.HUB_GENERATION_DEFAULTS and
    are passed explicitly by the trainer-level callers (training.gen_caption_from_loader_bart, utils/test_mmbart_clip_ddp.py).
    return_nbest: also return, per batch item, the finalized n-best list [(length-normalised score, ids)], best first — what
    `num_return_sequences=num_beams, output_scores=True` exposes as sequences / sequences_scores (the bookkeeping of ALL beams)."""
    cfg = model.config
    if model.arena is None:
        raise RuntimeError("call model.finalize(device) first")
    was_training = model.training
    if was_training:
        model.eval()                              # (walks ~1000 modules: 1 ms — skipped when the model already is in eval mode)
    eos, pad, start = cfg.eos_token_id, cfg.pad_token_id, cfg.decoder_start_token_id
    if forced_eos_token_id == "config":
        forced_eos_token_id = eos                     # BartConfig default forced_eos_token_id=2
    B = input_ids.shape[0]
    nb = num_beams
    R = B * nb
    dev = model.emb16_pad.device
    if encoded is not None:
        # the encoder side of this caption ran ahead on another stream (EncodeStage / CaptionPipeline): K/V and mask are staged
        if B != 1 or encoded.nb != nb:
            raise ValueError("generate(encoded=...): staged encoding is for one caption and the same beam count")
        enc_h, mask_u8, S, d = None, None, encoded.S, encoded.d
    else:
        mask_u8 = attention_mask.to(torch.uint8) if attention_mask.dtype != torch.uint8 else attention_mask
        enc_h = _run_encoder(model, use_graphs, add_ner_ffn, input_ids, mask_u8, image_features, name_ids, name_mask, face_features, face_mask)
        S, d = enc_h.shape[1], enc_h.shape[2]
    device_beams = bool(device_beams) and nb <= 16 and 2 * nb * nb <= 128 and max_length <= 512
    key = (R, S, max_length, nb, no_repeat_ngram_size, min_length, forced_eos_token_id, forced_bos_token_id,
           (float(length_penalty), bool(early_stopping)) if device_beams else None)
    sessions = model.__dict__.setdefault("_decode_sessions", {})
    ses = sessions.get(key)
    if ses is None:
        ses = sessions[key] = DecodeSession(model, R, S, max_length, nb, no_repeat_ngram_size, min_length, forced_eos_token_id, eos,
                                            forced_bos=forced_bos_token_id)
        if device_beams:
            ses.enable_device_beams(B, length_penalty, early_stopping, pad)
    if encoded is not None:
        ses.dec.begin_staged(encoded)
    else:
        ses.dec.begin(enc_h, mask_u8, nb)
    if _after_begin is not None:
        _after_begin()                                 # CaptionPipeline: the next caption's encoder side goes out before this beam search
    if device_beams:
        # on-device bookkeeping: the positions are enqueued back to back; ONE device->host copy of the final state
        try:
            cur_len, seqs_t, scores_t, done_t, hcnt, hscore, hlen, hseq = ses.run_device(start, use_graphs)
        except Exception:
            sessions.pop(key, None)                      # its graphs / buffers may hold a half-finished caption
            if was_training:
                model.train()
            raise
        ses.captions += 1
        out, nbest = [], []
        for b in range(B):
            hy = _BeamHyps(nb, length_penalty, early_stopping)
            for i in range(int(hcnt[b])):
                hy.beams.append((float(hscore[b, i]), hseq[b, i, :int(hlen[b, i])].tolist()))
            if hy.beams:
                hy.worst = min(sc for sc, _ in hy.beams)
            if not bool(done_t[b]):
                for j in range(nb):                       # BeamSearchScorer.finalize: open beams join the n-best list
                    hy.add(seqs_t[b * nb + j, :cur_len].tolist(), float(scores_t[b * nb + j]))
            ranked = sorted(hy.beams, key=lambda x: x[0])
            out.append(ranked[-1][1])
            nbest.append([(float(sc), list(sq)) for sc, sq in reversed(ranked)])
        L = min(max(len(o) for o in out) + 1, max_length)
        res = torch.full((B, L), pad, dtype=torch.long)
        for b, o in enumerate(out):
            res[b, :len(o)] = torch.tensor(o)
            if len(o) < L:
                res[b, len(o)] = eos
        if was_training:
            model.train()
        return (res.to(dev), nbest) if return_nbest else res.to(dev)

    seqs = [[start] for _ in range(R)]
    beam_scores = [0.0 if (r % nb) == 0 else -1e9 for r in range(R)]
    hyps = [_BeamHyps(nb, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    cur_len = 1
    Kc = 2 * nb
    new_src = list(range(R))
    while True:
        t = cur_len - 1
        bans = [_banned(s_, no_repeat_ngram_size) for s_ in seqs] if no_repeat_ngram_size > 0 else None
        try:
            tv, ti = ses.step(t, [s_[-1] for s_ in seqs], beam_scores, new_src, bans, use_graphs)
        except Exception:
            sessions.pop(key, None)
            if was_training:
                model.train()
            raise
        tv, ti = tv.tolist(), ti.tolist()
        new_seqs, new_scores, new_src = [], [], []
        for b in range(B):
            if done[b]:
                new_seqs += [seqs[b * nb] + [pad]] * nb; new_scores += [0.0] * nb; new_src += [b * nb] * nb
                continue
            # merge the per-beam top-2k lists into the group's top-2k (HF: topk over the [nb*V] scores of the group)
            cand = []
            for j in range(nb):
                row_v, row_i = tv[b * nb + j], ti[b * nb + j]
                for c in range(len(row_v)):
                    if row_i[c] >= 0:
                        cand.append((row_v[c], j, row_i[c]))
            cand.sort(key=lambda x: (-x[0], x[1] * model.V + x[2]))
            cand = cand[:Kc]
            chosen = []
            for rank, (sc, j, tok) in enumerate(cand):
                src = b * nb + j
                if tok == eos:
                    if rank >= nb:
                        continue
                    hyps[b].add(seqs[src], sc)
                else:
                    chosen.append((sc, tok, src))
                if len(chosen) == nb:
                    break
            done[b] = done[b] or hyps[b].is_done(cand[0][0], cur_len)
            while len(chosen) < nb:                                     # cannot happen with 2*nb candidates; keep shapes sane
                chosen.append((-1e9, pad, b * nb))
            for sc, tok, src in chosen:
                new_seqs.append(seqs[src] + [tok]); new_scores.append(sc); new_src.append(src)
        seqs = new_seqs
        beam_scores = new_scores
        cur_len += 1
        if all(done) or cur_len >= max_length:
            break
    ses.captions += 1
    out, nbest = [], []
    for b in range(B):
        if not done[b]:
            for j in range(nb):
                hyps[b].add(seqs[b * nb + j], float(beam_scores[b * nb + j]))
        ranked = sorted(hyps[b].beams, key=lambda x: x[0])
        out.append(ranked[-1][1])
        nbest.append([(float(sc), list(sq)) for sc, sq in reversed(ranked)])
    L = min(max(len(o) for o in out) + 1, max_length)
    res = torch.full((B, L), pad, dtype=torch.long)
    for b, o in enumerate(out):
        res[b, :len(o)] = torch.tensor(o)
        if len(o) < L:
            res[b, len(o)] = eos
    if was_training:
        model.train()
    return (res.to(dev), nbest) if return_nbest else res.to(dev)
