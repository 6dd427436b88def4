"""Per-kernel parity on the MI355X: each hand-written HIP kernel (called through the C-ABI) against a
plain PyTorch fp32 reference of the same op on the same seeded inputs.  Tolerances are bf16-level and
written next to each check."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FMIN = torch.finfo(torch.float32).min


@pytest.fixture(scope="module")
def K():
    from vacnic_amd import kernels
    return kernels


def rnd(*shape, scale=1.0, dtype=torch.bfloat16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)


def close(a, b, rtol, atol, what=""):
    a = a.float(); b = b.float()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    bad = (err > bound).sum().item()
    assert bad == 0, f"{what}: {bad}/{a.numel()} off; max err {err.max().item():.4g} (ref max {b.abs().max().item():.4g})"


# ---------------------------------------------------------------------------------------------- probes
def test_probe_layouts(K):
    out = K.probe_layouts().cpu().numpy()
    fa = lambda i, k: float((i * 3 + k) % 7 - 3)
    fb = lambda k, j: float((k * 5 + j * 2) % 9 - 4)
    A16 = np.array([[fa(i, k) for k in range(32)] for i in range(16)]); B16 = np.array([[fb(k, j) for j in range(16)] for k in range(32)])
    np.testing.assert_array_equal(out[:256].reshape(16, 16), A16 @ B16, err_msg="16x16x32 lane map")
    A32 = np.array([[fa(i, k) for k in range(16)] for i in range(32)]); B32 = np.array([[fb(k, j) for j in range(32)] for k in range(16)])
    C32 = A32 @ B32
    np.testing.assert_array_equal(out[256:1280].reshape(32, 32), C32, err_msg="32x32x16 lane map")
    tr = out[1280:1536].reshape(64, 4)
    for l in range(64):
        g, i = l >> 4, l & 15
        for e in range(4):
            assert tr[l, e] == (g * 4 + e) * 16 + i, f"tr-read lane {l} elem {e}: got {tr[l, e]}"
    oob = out[1536:1600]
    np.testing.assert_array_equal(oob[:32], np.arange(32) * 4 + 1.0)
    np.testing.assert_array_equal(oob[32:], np.zeros(32), err_msg="out-of-range LDS-DMA lanes must write 0")
    A2 = np.array([[float((i + 2 * k) % 5 - 2) for k in range(32)] for i in range(32)])
    np.testing.assert_array_equal(out[1600:2624].reshape(32, 32), A2 @ C32, err_msg="accumulator-as-B-operand k order")


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K_", [(256, 256, 128), (300, 200, 72), (2048, 1024, 1024), (64, 20, 80), (4, 2048, 512), (130, 50267, 64)])
def test_gemm_forward(K, M, N, K_):
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.05, seed=2); b = rnd(N, dtype=torch.float32, seed=3)
    out = K.gemm(x, w, M, N, K_, bias=b)
    ref = x.float() @ w.float().t() + b
    close(out, ref, 1e-2, 2e-2 * math.sqrt(K_ / 64) * 0.3, f"gemm NT {M}x{N}x{K_}")


def test_gemm_padded_ldo_and_f32_out(K):
    M, N, K_ = 100, 50267, 64
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.05, seed=2)
    ld = 50272
    out = torch.full((M, ld), 7.0, device="cuda", dtype=torch.bfloat16)
    K.gemm(x, w, M, N, K_, out=out, ldo=ld)
    ref = x.float() @ w.float().t()
    close(out[:, :N], ref, 1e-2, 1e-2, "padded ldo")
    assert (out[:, N:] == 7.0).all(), "columns >= N must not be written"
    o32 = K.gemm(x, w, M, N, K_, out_mode=1)
    close(o32, ref, 1e-4, 1e-4, "f32 out")


@pytest.mark.parametrize("act", ["gelu", "tanh", "quick_gelu"])
def test_gemm_epilogues(K, act):
    M, N, K_ = 200, 264, 128
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3)
    res = rnd(M, N, seed=4)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = K.gemm(x, w, M, N, K_, bias=b, act=act, preact=pre, residual=res)
    u = x.float() @ w.float().t() + b
    f = {"gelu": torch.nn.functional.gelu, "tanh": torch.tanh, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act]
    close(pre, u, 1e-2, 1e-2, "preact")
    close(out, f(u) + res.float(), 1e-2, 2e-2, f"act {act} + residual")
    # fused activation backward: out = (dy @ w2) * act'(u)
    dy = rnd(M, 96, seed=5); w2 = rnd(96, N, scale=0.1, seed=6)       # Linear(N -> 96): weight [96, N]
    du = K.gemm(dy, w2, M, N, 96, w_kstrided=True, act=act, dact_src=pre)
    uu = pre.float().requires_grad_(True)
    f(uu).backward(dy.float() @ w2.float())
    close(du, uu.grad, 2e-2, 2e-2, f"dact {act}")


@pytest.mark.parametrize("M,N,K_", [(256, 128, 256), (333, 264, 200), (2048, 1024, 4096)])
def test_gemm_dgrad_wkstrided(K, M, N, K_):
    # dX[M,N] = dY[M,K_] @ W[K_,N]   (W is an nn.Linear weight [out=K_, in=N])
    dy = rnd(M, K_, seed=1); w = rnd(K_, N, scale=0.05, seed=2)
    out = K.gemm(dy, w, M, N, K_, w_kstrided=True)
    close(out, dy.float() @ w.float(), 1e-2, 3e-2 * math.sqrt(K_ / 256), "dgrad")


@pytest.mark.parametrize("M,N,K_,split", [(128, 128, 256, 1), (264, 200, 1000, 1), (1024, 1024, 4096, 4), (50267, 64, 512, 2)])
def test_gemm_wgrad_both_kstrided(K, M, N, K_, split):
    # dW[M=out_features, N=in_features] += dY[K_, M]^T @ X[K_, N]
    ldy = (M + 7) // 8 * 8
    dyf = rnd(K_, ldy, seed=1); dyf[:, M:] = 0
    x = rnd(K_, N, seed=2)
    acc = torch.ones(M, N, device="cuda", dtype=torch.float32)
    K.gemm(dyf, x, M, N, K_, out=acc, ldx=ldy, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=split, alpha=0.5)
    ref = 1.0 + 0.5 * (dyf[:, :M].float().t() @ x.float())
    close(acc, ref, 2e-3, 2e-2 * math.sqrt(K_ / 256), "wgrad accumulate")


def test_gemm_row_split_dispatch(K):
    """auto dispatch cuts a small row remainder off into its own launch (ViT: M = 32*257 = 8224 = 8192 + 32); all
    row-indexed operands (out, preact, residual, dact_src, K-strided X) must be offset consistently."""
    M, N, K_ = 8224, 1024, 128
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3); res = rnd(M, N, seed=4)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = K.gemm(x, w, M, N, K_, bias=b, act="quick_gelu", preact=pre, residual=res)
    u = x.float() @ w.float().t() + b
    close(pre, u, 1e-2, 1e-2, "preact across the split"); close(out, u * torch.sigmoid(1.702 * u) + res.float(), 1e-2, 2e-2, "out across the split")
    d = K.gemm(x, w, M, N, K_, act="gelu", dact_src=pre)
    uu = pre.float().requires_grad_(True); torch.nn.functional.gelu(uu).backward(x.float() @ w.float().t())
    close(d, uu.grad, 2e-2, 2e-2, "dact across the split")
    xt = x.t().contiguous()                                     # [K_, M] K-strided X
    o32 = K.gemm(xt, w, M, N, K_, x_kstrided=True, out_mode=1)
    close(o32, x.float() @ w.float().t(), 1e-3, 1e-3, "K-strided X, f32 out across the split")


@pytest.mark.parametrize("hint", [64, 128, 256, 260, 261, 262, 264])
@pytest.mark.parametrize("M,N,K_", [(300, 520, 200), (1024, 768, 512), (257, 255 + 1, 64)])
def test_gemm_all_layouts_both_tile_configs(K, hint, M, N, K_):
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3)
    ref = x.float() @ w.float().t()
    tol = 2e-2 * math.sqrt(K_ / 64)
    close(K.gemm(x, w, M, N, K_, bias=b, tile_hint=hint), ref + b, 1e-2, tol, f"NN hint {hint}")
    wt = w.t().contiguous()                                   # [K_, N]: reduction-strided W
    close(K.gemm(x, wt, M, N, K_, w_kstrided=True, tile_hint=hint), ref, 1e-2, tol, f"NT hint {hint}")
    xt = torch.zeros(K_, (M + 7) // 8 * 8, device="cuda", dtype=torch.bfloat16); xt[:, :M] = x.t()
    for split in (1, 2):
        acc = torch.full((M, N), 2.0, device="cuda")
        K.gemm(xt, wt, M, N, K_, out=acc, ldx=xt.shape[1], x_kstrided=True, w_kstrided=True, out_mode=2, split_k=split, tile_hint=hint)
        close(acc, 2.0 + ref, 2e-3, tol, f"TT hint {hint} split {split}")
    close(K.gemm(xt, w, M, N, K_, ldx=xt.shape[1], x_kstrided=True, tile_hint=hint), ref, 1e-2, tol, f"TN hint {hint}")


@pytest.mark.parametrize("M,N,K_", [(16384, 1024, 1024), (16384, 4096, 1024)])
def test_gemm_target_shapes_elementwise(K, M, N, K_):
    """BASELINE target shapes at full M (the 'BART cross-attn GEMM' M=B*S=16384, N=K=1024 and the FFN up-projection) —
    every element against an fp32 product, in all three layouts of the step (forward NN, dgrad NT, wgrad TT)."""
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.05, seed=2); b = rnd(N, dtype=torch.float32, seed=3)
    ref = x.float() @ w.float().t()
    tol = 2e-2 * math.sqrt(K_ / 64) * 0.3
    close(K.gemm(x, w, M, N, K_, bias=b), ref + b, 1e-2, tol, "NN")
    wt = w.t().contiguous()
    close(K.gemm(x, wt, M, N, K_, w_kstrided=True), ref, 1e-2, tol, "NT")
    del ref
    # wgrad: dW[N, K_] = dY[M, N]^T X[M, K_]  (reduction over the 16384 rows, split-K atomics into an fp32 arena view)
    dy = rnd(M, N, scale=0.1, seed=4)
    acc = torch.zeros(N, K_, device="cuda")
    bsum = torch.zeros(N, device="cuda")
    K.gemm(dy, x, N, K_, M, out=acc, ldx=N, ldw=K_, ldo=K_, x_kstrided=True, w_kstrided=True, out_mode=2,
           split_k=K.wgrad_split(M, ((N + 127) // 128) * ((K_ + 127) // 128)), xsum=bsum)
    close(acc, dy.float().t() @ x.float(), 2e-3, 2e-2 * math.sqrt(M / 256), "TT")
    close(bsum, dy.float().sum(0), 2e-3, 2e-2 * math.sqrt(M / 256), "TT fused bias gradient")


def test_wgrad_group_matches_torch_and_is_reproducible(K):
    """vacnic_wgrad_group: several weight gradients dW += dY^T X (+ bias gradients) in one launch, no split-K — full-size
    encoder shapes (d x d, 3d x d, d x 4d at M = 16384), decoder-sized reductions, ragged N / K / M, strided operands, one job
    without a bias, accumulation into a non-zero dW, more than 16 output blocks in one call; every element against an fp32
    product, and two runs bit-identical (one writer per element, fixed summation order)."""
    from vacnic_amd import _lib
    shapes = [(16384, 1024, 1024, True), (16384, 3072, 1024, True), (16384, 1024, 4096, True), (2048, 1024, 1024, True),
              (2048, 4096, 1024, False), (1030, 520, 648, True), (1536, 1152, 2048, True)]
    jobs, refs = [], []
    for i, (M, N, K_, has_bias) in enumerate(shapes):
        ldy, ldx = N + (8 if i % 2 else 0), K_ + (16 if i % 3 == 0 else 0)
        dyb = rnd(M, ldy, scale=0.1, seed=10 + i); xb = rnd(M, ldx, seed=30 + i)
        dy, x = dyb[:, :N], xb[:, :K_]
        dw0 = rnd(N, K_, dtype=torch.float32, seed=50 + i)
        db0 = rnd(N, dtype=torch.float32, seed=70 + i) if has_bias else None
        jobs.append((dy, x, dw0, db0))
        refs.append((dw0.clone() + dy.float().t() @ x.float(), (db0.clone() + dy.float().sum(0)) if has_bias else None, M))
    keep = [(j[2].clone(), j[3].clone() if j[3] is not None else None) for j in jobs]
    K.wgrad_group(jobs)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb, M) in zip(jobs, refs):
        close(dw, rw, 2e-3, 2e-2 * math.sqrt(M / 256), f"dW {tuple(dw.shape)} M={M}")
        if db is not None:
            close(db, rb, 2e-3, 2e-2 * math.sqrt(M / 256), f"dbias {tuple(db.shape)} M={M}")
    first = [j[2].clone() for j in jobs]
    for j, (w0, b0) in zip(jobs, keep):
        j[2].copy_(w0)
        if b0 is not None:
            j[3].copy_(b0)
    K.wgrad_group(jobs)
    torch.cuda.synchronize()
    for j, f in zip(jobs, first):
        assert torch.equal(j[2], f), "weight gradients must be bitwise reproducible"
    # argument checks: misaligned rows are refused
    bad = (rnd(1024, 1028, seed=1)[:, :1024], jobs[0][1][:1024], torch.zeros(1024, 1024, device="cuda"), None)
    with pytest.raises(ValueError):
        K.wgrad_group([bad])


@pytest.mark.parametrize("hint", [256, 264, 256 + 64000, 256 + 16000])
def test_gemm_persistent_many_tiles_and_ragged_rows(K, hint):
    """more tiles than CUs (the persistent work loop wraps: 65 x 5 = 325 tiles of 256x256) with a ragged last row block;
    hint + 64000 = one workgroup per tile, + 16000 = fp32 staged epilogue: all must agree with the fp32 product."""
    M, N, K_ = 16384 + 100, 1280, 192
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3); res = rnd(M, N, seed=4)
    ref = x.float() @ w.float().t() + b
    out = torch.full((M + 8, N), 7.0, device="cuda", dtype=torch.bfloat16)
    K.gemm(x, w, M, N, K_, bias=b, out=out, tile_hint=hint)
    close(out[:M], ref, 1e-2, 2e-2, f"plain hint {hint}")
    assert (out[M:] == 7.0).all(), "rows >= M must not be written"
    close(K.gemm(x, w, M, N, K_, bias=b, residual=res, tile_hint=hint), ref + res.float(), 1e-2, 3e-2, f"residual hint {hint}")


@pytest.mark.parametrize("act", ["gelu", "tanh", "quick_gelu"])
@pytest.mark.parametrize("hint", [256, 264])
def test_gemm_epilogues_bf16_staged_256_row_tiles(K, act, hint):
    """the single-pass bf16 epilogue of the 256-row tiles: activation, saved pre-activation, residual, fused activation
    backward (on the saved pre-activation), strided output rows."""
    M, N, K_ = 700, 520, 128
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3)
    res = rnd(M, N, seed=4)
    f = {"gelu": torch.nn.functional.gelu, "tanh": torch.tanh, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act]
    u = x.float() @ w.float().t() + b
    close(K.gemm(x, w, M, N, K_, bias=b, act=act, tile_hint=hint), f(u), 1e-2, 1e-2, "act only (in registers)")
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ld = 528
    out = torch.full((M, ld), 3.0, device="cuda", dtype=torch.bfloat16)
    out2 = K.gemm(x, w, M, N, K_, bias=b, act=act, preact=pre, tile_hint=hint)             # bf16 path, saved pre-activation
    close(pre, u, 1e-2, 1e-2, "preact")
    close(out2, f(pre.float()), 1e-2, 1e-2, f"act {act} on the saved pre-activation")
    close(out2, f(u), 1.5e-2, 2e-2, f"act {act} vs fp32")
    close(K.gemm(x, w, M, N, K_, bias=b, act=act, residual=res, tile_hint=hint), f(u) + res.float(), 1e-2, 2e-2, "act + residual")
    pre2 = torch.empty_like(pre)                                                             # two extras: fp32-staged path
    out3 = K.gemm(x, w, M, N, K_, bias=b, act=act, preact=pre2, residual=res, tile_hint=hint)
    close(pre2, u, 1e-2, 1e-2, "preact (with residual)")
    close(out3, f(u) + res.float(), 1e-2, 2e-2, f"act {act} + preact + residual")
    K.gemm(x, w, M, N, K_, bias=b, out=out, ldo=ld, tile_hint=hint)
    close(out[:, :N], u, 1e-2, 1e-2, "padded ldo"); assert (out[:, N:] == 3.0).all()
    dy = rnd(M, 96, seed=5); w2 = rnd(96, N, scale=0.1, seed=6)
    du = K.gemm(dy, w2, M, N, 96, w_kstrided=True, act=act, dact_src=pre, tile_hint=hint)
    uu = pre.float().requires_grad_(True)
    f(uu).backward(dy.float() @ w2.float())
    close(du, uu.grad, 2e-2, 2e-2, f"dact {act}")


@pytest.mark.parametrize("hint", [0, 64, 128, 256, 264, 261])
@pytest.mark.parametrize("M,N,K_,split", [(1024, 1024, 2048, 1), (1024, 768, 4096, 4), (200, 520, 1000, 2), (80, 80, 4096, 8)])
def test_gemm_wgrad_fused_bias_gradient(K, hint, M, N, K_, split):
    """xsum: column sums of dY (the nn.Linear bias gradient) out of the weight-gradient GEMM's own fragments."""
    ldy = (M + 7) // 8 * 8
    dy = rnd(K_, ldy, seed=1); dy[:, M:] = 0
    x = rnd(K_, N, seed=2)
    acc = torch.ones(M, N, device="cuda")
    bs = torch.full((M,), 0.5, device="cuda")
    K.gemm(dy, x, M, N, K_, out=acc, ldx=ldy, x_kstrided=True, w_kstrided=True, out_mode=2, split_k=split, xsum=bs, tile_hint=hint)
    close(acc, 1.0 + dy[:, :M].float().t() @ x.float(), 2e-3, 2e-2 * math.sqrt(K_ / 256), "wgrad")
    close(bs, 0.5 + dy[:, :M].float().sum(0), 2e-3, 2e-2 * math.sqrt(K_ / 256), f"bias gradient hint {hint} split {split}")


@pytest.mark.parametrize("hint", [64, 128, 256, 264])
@pytest.mark.parametrize("M,N,K_,split", [(2048, 1024, 4096, 4), (300, 520, 1000, 2), (640, 1024, 3072, 8), (257, 256, 512, 3)])
def test_gemm_split_k_ordered_fixup(K, hint, M, N, K_, split):
    """split_k > 1 through the ordered fix-up (workspace + arrival tickets) instead of fp32 atomics: every layout, bf16 output with
    bias + activation + residual (impossible with atomics), fp32 store, accumulate; bitwise reproducible; counters left at zero."""
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3); res = rnd(M, N, seed=4)
    ref = x.float() @ w.float().t()
    tol = 2e-2 * math.sqrt(K_ / 64)
    kw = dict(split_k=split, fixup=True, tile_hint=hint)
    o1 = K.gemm(x, w, M, N, K_, bias=b, act="gelu", residual=res, **kw)
    close(o1, torch.nn.functional.gelu(ref + b) + res.float(), 1e-2, tol, f"NN bf16 + bias + gelu + residual, hint {hint} split {split}")
    for _ in range(3):                                           # who arrives last varies; the sum order does not
        assert torch.equal(o1, K.gemm(x, w, M, N, K_, bias=b, act="gelu", residual=res, **kw)), "fix-up must be bitwise reproducible"
    wt = w.t().contiguous()
    close(K.gemm(x, wt, M, N, K_, w_kstrided=True, out_mode=1, **kw), ref, 2e-3, tol, "NT f32 store")
    # the unsplit launch sums the same products in another order: equal to fp32 round-off, and both equal the reference
    close(K.gemm(x, wt, M, N, K_, w_kstrided=True, out_mode=1, **kw), K.gemm(x, wt, M, N, K_, w_kstrided=True, out_mode=1, tile_hint=hint), 1e-4, 1e-3, "split vs unsplit")
    xt = torch.zeros(K_, (M + 7) // 8 * 8, device="cuda", dtype=torch.bfloat16); xt[:, :M] = x.t()
    bs0 = torch.full((M,), 0.5, device="cuda")
    accs = []
    for _ in range(2):
        acc = torch.full((M, N), 2.0, device="cuda"); bs = bs0.clone()
        K.gemm(xt, wt, M, N, K_, out=acc, ldx=xt.shape[1], x_kstrided=True, w_kstrided=True, out_mode=2, xsum=bs, **kw)
        accs.append(acc)
    close(accs[0], 2.0 + ref, 2e-3, tol, "TT accumulate (weight-gradient layout)")
    close(bs, 0.5 + x.float().sum(1), 2e-3, tol, "TT fused bias gradient beside the fix-up")
    assert torch.equal(accs[0], accs[1]), "weight gradient through the fix-up: bitwise reproducible"
    torch.cuda.synchronize()
    for ws, cnt in K._FIX.values():
        assert int(cnt.abs().sum()) == 0, "arrival counters must be zero after every launch"


# ------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, key_mask, causal, scale):
    B, Tq, H, hd = q.shape
    Tk = k.shape[1]
    s = torch.einsum("bqhd,bkhd->bhqk", q.float() * scale, k.float())
    if key_mask is not None:
        s = s + ((1.0 - key_mask.float()) * FMIN)[:, None, None, :]
    if causal:
        cm = torch.full((Tq, Tk), FMIN, device=q.device).triu(1)
        s = s + cm
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v.float()).reshape(B, Tq, H * hd)


SHAPES = [  # B, H, Tq, Tk, masked, causal
    (2, 4, 128, 128, False, False), (2, 3, 200, 200, True, False), (2, 4, 96, 40, False, False),
    (3, 2, 80, 84, True, False), (2, 4, 64, 64, False, True), (2, 2, 64, 300, True, False),
    (1, 2, 257, 257, False, False), (2, 2, 20, 512, True, False), (1, 16, 512, 512, True, False),
    (1, 4, 1024, 1024, True, False), (1, 2, 64, 1024, True, False), (2, 2, 50, 50, False, False)]


@pytest.mark.parametrize("B,H,Tq,Tk,masked,causal", SHAPES)
def test_attention_fwd_bwd(K, B, H, Tq, Tk, masked, causal):
    d = H * 64
    # q from a fused [B,Tq,3d] buffer (stride 3d), k/v from a [B,Tk,2d] buffer: exercises strided views
    qkv = rnd(B, Tq, 3 * d, seed=1); kv = rnd(B, Tk, 2 * d, seed=2)
    q = qkv[..., 2 * d:]; k = kv[..., :d]; v = kv[..., d:]
    mask = None
    if masked:
        lens = torch.randint(max(1, Tk // 4), Tk + 1, (B,), generator=torch.Generator().manual_seed(3))
        mask = (torch.arange(Tk)[None, :] < lens[:, None]).to(torch.uint8).cuda()
    out, lse = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=mask, causal=causal, scale=0.125)
    qf = q.float().reshape(B, Tq, H, 64).requires_grad_(True)
    kf = k.float().reshape(B, Tk, H, 64).requires_grad_(True)
    vf = v.float().reshape(B, Tk, H, 64).requires_grad_(True)
    ref = attn_ref(qf, kf, vf, mask, causal, 0.125)
    close(out, ref, 2e-2, 2e-2, "attn fwd")
    dout = rnd(B, Tq, d, seed=4)
    ref.backward(dout.float())
    dqkv = torch.zeros_like(qkv); dkv = torch.zeros_like(kv)
    K.attn_bwd(q, k, v, out, dout, lse, dqkv[..., 2 * d:], dkv[..., :d], dkv[..., d:], B, H, Tq, Tk, key_mask=mask,
               causal=causal, scale=0.125)
    s = max(1.0, math.sqrt(Tk / 64))
    close(dqkv[..., 2 * d:], qf.grad.reshape(B, Tq, d), 3e-2, 3e-2 * s, "dq")
    close(dkv[..., :d], kf.grad.reshape(B, Tk, d), 3e-2, 3e-2 * s, "dk")
    close(dkv[..., d:], vf.grad.reshape(B, Tk, d), 3e-2, 3e-2 * s, "dv")
    assert (dqkv[..., :2 * d] == 0).all()


def test_attention_fully_masked_row_is_uniform(K):
    B, H, Tq, Tk = 1, 1, 32, 64
    q = rnd(B, Tq, 64, seed=1); k = rnd(B, Tk, 64, seed=2); v = rnd(B, Tk, 64, seed=3)
    mask = torch.zeros(B, Tk, dtype=torch.uint8, device="cuda")
    out, _ = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=mask)
    close(out, v.float().mean(1, keepdim=True).expand(B, Tq, 64), 1e-2, 1e-2, "all-masked row = uniform softmax like torch")


@pytest.mark.parametrize("B,H,Tk,masked", [(5, 16, 1, False), (5, 16, 37, False), (8, 12, 512, True), (3, 4, 600, True), (2, 2, 64, True)])
def test_attention_single_query_decode_path(K, B, H, Tk, masked):
    """Tq == 1 (KV-cached decoder step): one-wave-per-(row, head) kernel; strided cache views, key mask, lse, all-masked row."""
    d = H * 64
    q = rnd(B, 1, 3 * d, seed=1)[..., 2 * d:]
    cache = rnd(B, Tk + 3, 2 * d, seed=2)                       # cache longer than the valid prefix
    k = cache[:, :Tk, :d]; v = cache[:, :Tk, d:]
    mask = None
    if masked:
        lens = torch.randint(1, Tk + 1, (B,), generator=torch.Generator().manual_seed(3))
        lens[0] = 0                                              # fully masked row -> uniform softmax
        mask = (torch.arange(Tk)[None, :] < lens[:, None]).to(torch.uint8).cuda()
    out, lse = K.attn_fwd(q, k, v, B, H, 1, Tk, key_mask=mask, scale=0.125)
    ref = attn_ref(q.float().reshape(B, 1, H, 64), k.float().reshape(B, Tk, H, 64), v.float().reshape(B, Tk, H, 64), mask, False, 0.125)
    close(out, ref, 1e-2, 1e-2, "decode attention")
    s = torch.einsum("bqhd,bkhd->bhqk", q.float().reshape(B, 1, H, 64) * 0.125, k.float().reshape(B, Tk, H, 64))
    if mask is not None:
        s = s + ((1.0 - mask.float()) * FMIN)[:, None, None, :]
    close(lse, torch.logsumexp(s, -1), 1e-3, 1e-3 * max(1.0, 0.0), "decode lse") if mask is None else None


# ------------------------------------------------------------------------------------------------- LN
@pytest.mark.parametrize("R,D", [(64, 1024), (37, 768), (5, 512), (128, 2048)])
def test_add_ln(K, R, D):
    x = rnd(R, D, seed=1); res = rnd(R, D, seed=2)
    g = rnd(D, dtype=torch.float32, seed=3) * 0.1 + 1.0; b = rnd(D, dtype=torch.float32, seed=4) * 0.1
    out, mean, rstd = K.add_ln_fwd(x, res, g, b)
    xf = x.float().requires_grad_(True); rf = res.float().requires_grad_(True)
    gf = g.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf + rf, (D,), gf, bf, 1e-5)
    close(out, ref, 1e-2, 1e-2, "add_ln fwd")
    dout = rnd(R, D, seed=5)
    ref.backward(dout.float())
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dx, dres = K.add_ln_bwd(dout, x, res, g, mean, rstd, dg, db)
    close(dx, xf.grad, 2e-2, 2e-2, "add_ln dx")
    assert dres is dx
    close(dg, gf.grad, 1e-2, 1e-2 * math.sqrt(R), "dgamma"); close(db, bf.grad, 1e-2, 1e-2 * math.sqrt(R), "dbeta")
    # plain LN (no residual)
    out2, _, _ = K.add_ln_fwd(x, None, g, b)
    close(out2, torch.nn.functional.layer_norm(x.float(), (D,), g, b, 1e-5), 1e-2, 1e-2, "plain LN")


def _recover_attn_mask(K, q, k, B, H, Tq, Tk, p, seed, key_mask=None, causal=False):
    """(P o M / (1 - p_q)) of the kernel itself: with V = a one-hot window of 64 keys the output row of head h IS the dropped
    probability row restricted to that window."""
    pm = torch.zeros(B, H, Tq, Tk)
    for w0 in range(0, Tk, 64):
        v = torch.zeros(B, Tk, H * 64, device="cuda", dtype=torch.bfloat16)
        n = min(64, Tk - w0)
        for h in range(H):
            v[:, w0:w0 + n, h * 64:h * 64 + n] = torch.eye(n, device="cuda", dtype=torch.bfloat16)
        o, _ = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=key_mask, causal=causal, p_drop=p, seed=seed)
        pm[:, :, :, w0:w0 + n] = o.float().view(B, Tq, H, 64).permute(0, 2, 1, 3)[..., :n].cpu()
    return pm


@pytest.mark.parametrize("Tq,Tk,causal", [(64, 64, False), (100, 200, False), (72, 72, True)])
def test_attention_probability_dropout(K, Tq, Tk, causal):
    """attention_dropout > 0 (MFULL:546): P -> P o M / (1 - p) AFTER the softmax, mask from Philox(seed, b, h, q, k) regenerated by
    the two backward kernels.  The mask is read off the kernel (one-hot values), checked for its rate, independence of V and
    seed dependence; forward and all three gradients are then compared with torch autograd using that mask."""
    B, H, p, seed = 2, 2, 0.25, 777
    q = rnd(B, Tq, H * 64, scale=0.7, seed=1); k = rnd(B, Tk, H * 64, scale=0.7, seed=2); v = rnd(B, Tk, H * 64, seed=3)
    km = torch.ones(B, Tk, dtype=torch.uint8); km[1, Tk - 7:] = 0; km = km.cuda()
    pm = _recover_attn_mask(K, q, k, B, H, Tq, Tk, p, seed, key_mask=km, causal=causal)
    qf, kf, vf = (t.float().cpu().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    sc = qf @ kf.transpose(-1, -2) * 0.125
    sc = sc + (1 - km.cpu().float())[:, None, None, :] * torch.finfo(torch.float32).min
    if causal:
        sc = sc + torch.triu(torch.full((Tq, Tk), torch.finfo(torch.float32).min), 1)
    P = torch.softmax(sc, -1)
    big = P > 1e-3                                            # where a dropped entry is distinguishable from a tiny probability
    keep = pm > 0
    p_q = round(p * 256) / 256
    frac = keep[big].float().mean().item()
    assert abs(frac - (1 - p_q)) < 0.03, f"keep fraction {frac} vs {1 - p_q}"
    M = torch.where(keep | ~big, torch.full_like(P, 1 / (1 - p_q)), torch.zeros_like(P))
    assert ((pm - P * M).abs()[big] < 2e-2 * (P * M)[big] + 2e-3).all(), "kept probabilities are P / (1 - p)"
    # the kernel with real values == torch with that mask, forward and backward
    qa, ka, va = (t.clone().requires_grad_(True) for t in (qf, kf, vf))
    s2 = qa @ ka.transpose(-1, -2) * 0.125 + (1 - km.cpu().float())[:, None, None, :] * torch.finfo(torch.float32).min
    if causal:
        s2 = s2 + torch.triu(torch.full((Tq, Tk), torch.finfo(torch.float32).min), 1)
    want = (torch.softmax(s2, -1) * M) @ va
    out, lse = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=km, causal=causal, p_drop=p, seed=seed)
    close(out.view(B, Tq, H, 64).transpose(1, 2).cpu(), want, 2e-2, 2e-2, "dropout forward")
    close(lse.cpu(), torch.logsumexp(s2, -1), 1e-3, 1e-3, "lse is the undropped one")
    dout = rnd(B, Tq, H * 64, seed=9)
    want.backward(dout.float().cpu().view(B, Tq, H, 64).transpose(1, 2))
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    K.attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, key_mask=km, causal=causal, p_drop=p, seed=seed)
    for name, got, ref in (("dq", dq, qa.grad), ("dk", dk, ka.grad), ("dv", dv, va.grad)):
        g = got.float().cpu().view(B, -1, H, 64).transpose(1, 2)
        assert ((g - ref).norm() / ref.norm()).item() < 2e-2, (name, ((g - ref).norm() / ref.norm()).item())
    out2, _ = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=km, causal=causal, p_drop=p, seed=seed)
    assert torch.equal(out, out2), "same seed -> same mask"
    out3, _ = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=km, causal=causal, p_drop=p, seed=seed + 1)
    assert not torch.equal(out, out3)
    out0, _ = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=km, causal=causal, p_drop=0.0, seed=seed)
    outn, _ = K.attn_fwd(q, k, v, B, H, Tq, Tk, key_mask=km, causal=causal)
    assert torch.equal(out0, outn), "p = 0 is the plain kernel"


def test_activation_dropout_kernel_and_mlp2(K):
    """activation_dropout > 0 (MFULL:649,740,874): in-place Philox dropout of act(fc1 x); the same call on the gradient is its
    backward.  Kernel: rate, scaling, determinism; Mlp2Fn: forward / input gradient against torch with the recovered mask."""
    from vacnic_amd import ops
    x = rnd(512, 1024, seed=1)
    y = K.dropout_(x.clone(), 0.2, 99)
    keep = y.float() != 0
    assert abs(keep.float().mean().item() - 0.8) < 0.01
    close(y.float()[keep], (x.float() / 0.8)[keep], 1e-2, 1e-3, "kept values scaled by 1 / (1 - p)")
    assert torch.equal(y, K.dropout_(x.clone(), 0.2, 99)) and not torch.equal(y, K.dropout_(x.clone(), 0.2, 100))
    M, d, F = 96, 256, 512
    w1 = rnd(F, d, scale=0.05, seed=2); b1 = rnd(F, dtype=torch.float32, seed=3) * 0.1
    w2 = rnd(d, F, scale=0.05, seed=4); b2 = rnd(d, dtype=torch.float32, seed=5) * 0.1
    s1 = ops.LinearSpec(w1, b1, torch.zeros(F, d, device="cuda"), torch.zeros(F, device="cuda"))
    s2 = ops.LinearSpec(w2, b2, torch.zeros(d, F, device="cuda"), torch.zeros(d, device="cuda"))
    xin = rnd(M, d, seed=6).requires_grad_(True)
    anchor = torch.zeros(1, device="cuda", requires_grad=True)
    ops.Rng.device_counter().zero_()
    out = ops.Mlp2Fn.apply(xin, anchor, s1, s2, "gelu", True, False, 0.3, 4242)
    dout = rnd(M, d, seed=7)
    out.backward(dout)
    ops.flush_wgrads()
    # the mask of the hidden activations: the same kernel call on an array of ones
    mask = K.dropout_(torch.ones(M, F, device="cuda", dtype=torch.bfloat16), 0.3, 4242, ops.Rng.device_counter()).float()
    xr = xin.detach().float().requires_grad_(True)
    h = torch.nn.functional.gelu(xr @ w1.float().t() + b1) * mask
    ref = h @ w2.float().t() + b2
    close(out, ref, 2e-2, 2e-2, "mlp2 with activation dropout")
    ref.backward(dout.float())
    assert ((xin.grad.float() - xr.grad).norm() / xr.grad.norm()).item() < 2e-2
    assert ((s2.wgrad - (dout.float().t() @ h.detach())).norm() / (dout.float().t() @ h.detach()).norm()).item() < 2e-2, "fc2's weight gradient sees the dropped activations"


@pytest.mark.parametrize("hint", [64, 128, 256, 264])
@pytest.mark.parametrize("M,N,K_", [(700, 512, 128), (96, 80, 256), (2048, 4096, 1024)])
def test_gemm_fused_activation_dropout(K, hint, M, N, K_):
    """activation dropout inside the GEMM epilogues (vacnic_gemm_args.drop_p): the forward Linear applies the mask after the
    activation (with and without a saved pre-activation), the dgrad applies the SAME mask after act'; the mask is the one
    vacnic_dropout_bf16 produces for the same (seed, device counter).  Every tile configuration (128 runs on the 64-row tiles)."""
    from vacnic_amd import ops
    if hint in (256, 264) and N < 256:
        pytest.skip("256-row tiles need N >= 256")
    p, seed = 0.1, 777
    cnt = ops.Rng.device_counter(); cnt.fill_(5)
    x = rnd(M, K_, seed=1); w = rnd(N, K_, scale=0.1, seed=2); b = rnd(N, dtype=torch.float32, seed=3)
    mask = K.dropout_(torch.ones(M, N, device="cuda", dtype=torch.bfloat16), p, seed, cnt).float()
    assert abs((mask != 0).float().mean().item() - (1 - 26 / 256)) < 0.01 and torch.allclose(mask[mask != 0], torch.tensor(256.0 / 230.0, device="cuda"), rtol=4e-3)
    u = x.float() @ w.float().t() + b
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = K.gemm(x, w, M, N, K_, bias=b, act="gelu", preact=pre, drop=(p, seed, cnt), tile_hint=hint)
    close(pre, u, 1e-2, 1e-2, "saved pre-activation is not dropped")
    close(h, torch.nn.functional.gelu(pre.float()) * mask, 1e-2, 1e-2, f"gelu then dropout, hint {hint}")
    assert bool((h[mask == 0] == 0).all()), "dropped positions are exact zeros"
    h2 = K.gemm(x, w, M, N, K_, bias=b, act="gelu", drop=(p, seed, cnt), tile_hint=hint)             # no saved pre-activation (fp32-staged path)
    close(h2, torch.nn.functional.gelu(u) * mask, 1.5e-2, 2e-2, "gelu then dropout without preact")
    dy = rnd(M, 96, seed=5); w2 = rnd(96, N, scale=0.1, seed=6)
    du = K.gemm(dy, w2, M, N, 96, w_kstrided=True, act="gelu", dact_src=pre, drop=(p, seed, cnt), tile_hint=hint)
    uu = pre.float().requires_grad_(True)
    (torch.nn.functional.gelu(uu) * mask).backward(dy.float() @ w2.float())
    close(du, uu.grad, 2e-2, 2e-2, f"act' then the same mask, hint {hint}")
    cnt.fill_(6)                                                  # the device counter is part of the key: fresh mask every step
    assert not torch.equal(h, K.gemm(x, w, M, N, K_, bias=b, act="gelu", preact=pre, drop=(p, seed, cnt), tile_hint=hint))
    cnt.zero_()


def test_add_ln_dropout_consistency(K):
    R, D, p = 256, 1024, 0.1
    x = rnd(R, D, seed=1); res = torch.zeros(R, D, device="cuda", dtype=torch.bfloat16)
    g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    # recover the keep mask from the backward: dx = dh * keep/(1-p), dres = dh
    out, mean, rstd = K.add_ln_fwd(x, res, g, b, p_drop=p, seed=1234)
    dout = rnd(R, D, seed=5)
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dx, dres = K.add_ln_bwd(dout, x, res, g, mean, rstd, dg, db, p_drop=p, seed=1234)
    keep = (dx.float().abs() > 0) | (dres.float().abs() == 0)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, f"keep fraction {frac}"
    xd = torch.where(keep, x.float() / (1 - p), torch.zeros_like(x.float()))
    close(out, torch.nn.functional.layer_norm(xd, (D,), g, b, 1e-5), 2e-2, 2e-2, "LN(dropout(x)) with the recovered mask")
    out2, _, _ = K.add_ln_fwd(x, res, g, b, p_drop=p, seed=1234)
    assert torch.equal(out, out2), "same seed -> same mask"
    out3, _, _ = K.add_ln_fwd(x, res, g, b, p_drop=p, seed=99)
    assert not torch.equal(out, out3)


@pytest.mark.parametrize("D", [768, 1024])
def test_embed_ln(K, D):
    B, T, V = 3, 40, 1000
    gen = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (B, T), generator=gen).cuda(); ids[0, -5:] = 1
    emb = rnd(V, D, scale=0.5, seed=1); pos = rnd(T + 2, D, scale=0.5, seed=2)
    g = rnd(D, dtype=torch.float32, seed=3) * 0.1 + 1.0; b = rnd(D, dtype=torch.float32, seed=4) * 0.1
    out, mean, rstd = K.embed_ln_fwd(ids, emb, pos, g, b, embed_scale=1.0)
    ef = emb.float().requires_grad_(True); pf = pos.float().requires_grad_(True)
    gf = g.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    h = torch.nn.functional.embedding(ids, ef, padding_idx=1) + pf[2:2 + T][None]
    ref = torch.nn.functional.layer_norm(h, (D,), gf, bf, 1e-5)
    close(out, ref, 1e-2, 1e-2, "embed_ln fwd")
    dout = rnd(B, T, D, seed=5)
    ref.backward(dout.float())
    de = torch.zeros(V, D, device="cuda"); dp = torch.zeros(T + 2, D, device="cuda")
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    K.embed_ln_bwd(ids, emb, pos, dout, g, mean, rstd, de, dp, dg, db, padding_idx=1)
    close(de, ef.grad, 2e-2, 2e-2, "dembed (padding_idx row gets no grad)")
    close(dp, pf.grad, 2e-2, 3e-2, "dpos"); close(dg, gf.grad, 2e-2, 0.1, "dgamma"); close(db, bf.grad, 2e-2, 0.1, "dbeta")


def test_name_embed_mean(K):
    B, Nn, Ln, V, D = 2, 5, 8, 500, 1024
    ids = torch.randint(0, V, (B, Nn, Ln), generator=torch.Generator().manual_seed(0)).cuda()
    emb = rnd(V, D, scale=0.5, seed=1); pos = rnd(Ln + 2, D, scale=0.5, seed=2)
    g = rnd(D, dtype=torch.float32, seed=3) * 0.1 + 1.0; b = rnd(D, dtype=torch.float32, seed=4) * 0.1
    out = K.name_embed_mean(ids, emb, pos, g, b)
    h = emb.float()[ids] + pos.float()[2:2 + Ln][None, None]
    ref = torch.nn.functional.layer_norm(h, (D,), g, b, 1e-5).mean(2)
    close(out, ref, 1e-3, 1e-3, "name embed mean")


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("f32", [False, True])
def test_cross_entropy(K, f32):
    R, V, ld = 50, 50267, 50272
    logits = torch.zeros(R, ld, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    logits[:, :V] = rnd(R, V, scale=2.0, seed=1).to(logits.dtype)
    logits[:, V:] = 100.0                      # padding columns must be ignored
    tgt = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2)).cuda(); tgt[::7] = 1
    lse, acc = K.ce_fwd(logits, tgt, V, ignore_index=1)
    lf = logits[:, :V].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, tgt, ignore_index=1)
    got = (acc[0] / acc[1]).item()
    assert abs(got - ref.item()) < 2e-3 * abs(ref.item()), (got, ref.item())
    assert acc[1].item() == (tgt != 1).sum().item()
    ref.backward()
    dl = torch.empty(R, ld, device="cuda", dtype=torch.bfloat16)
    K.ce_bwd(logits, tgt, V, lse, acc, dl)
    close(dl[:, :V], lf.grad, 2e-2, 1e-5, "dlogits")
    assert (dl[:, V:] == 0).all()


@pytest.mark.parametrize("R,V,d", [(300, 50267, 256), (64, 1000, 128), (2048, 50267, 1024)])
def test_lmhead_ce_fused_matches_materialised_logits(K, R, V, d):
    """lm_head + CrossEntropyLoss(ignore_index=1) without the [R, V] logits: forward statistics, every dlogits chunk and the
    autograd Function's dh / dE against torch on materialised fp32 logits."""
    from vacnic_amd import ops
    Vp = (V + 31) // 32 * 32
    h = rnd(R, d, scale=1.0, seed=1)
    E = torch.zeros(Vp, d, device="cuda", dtype=torch.bfloat16)
    E[:V] = rnd(V, d, scale=0.08, seed=2)
    tgt = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(3)).cuda(); tgt[::7] = 1; tgt[5] = V - 1; tgt[6] = 0
    hf = h.float().requires_grad_(True); Ef = E[:V].float().requires_grad_(True)
    logits = hf @ Ef.t()
    ref = torch.nn.functional.cross_entropy(logits, tgt, ignore_index=1)
    ref.backward()
    lse, acc = K.lmhead_ce_fwd(h, E, tgt, V, ignore_index=1)
    close(lse, torch.logsumexp(logits.detach(), -1), 1e-4, 2e-3, "row lse")
    assert acc[1].item() == (tgt != 1).sum().item()
    assert abs((acc[0] / acc[1]).item() - ref.item()) <= 2e-3 * abs(ref.item()), ((acc[0] / acc[1]).item(), ref.item())
    # dlogits, chunk by chunk (ragged last chunk, zero pad columns)
    rowp = K.lmhead_ce_rowp(lse, tgt, acc, grad_out=None, grad_scale=1.0, ignore_index=1)
    want = (torch.softmax(logits.detach(), -1) - torch.nn.functional.one_hot(tgt, V).float()) * ((tgt != 1).float() / acc[1])[:, None]
    CH = 16384
    dl = torch.full((R, CH), 9.0, device="cuda", dtype=torch.bfloat16)
    for c0 in range(0, V, CH):
        n = min(CH, V - c0); n8 = (n + 7) // 8 * 8
        K.lmhead_ce_dlogits(h, E, tgt, V, rowp, dl, c0, n, ignore_index=1)
        close(dl[:, :n], want[:, c0:c0 + n], 2e-2, 2e-6, f"dlogits chunk at {c0}")
        assert (dl[:, n:n8] == 0).all(), "pad columns of a ragged chunk must be zeros"
    # the autograd Function end to end (dh, dE accumulated into a gradient view)
    egrad = torch.zeros(Vp, d, device="cuda")
    hh = h.clone().requires_grad_(True)
    anchor = torch.zeros(1, device="cuda", requires_grad=True)
    loss, _ = ops.lm_head_ce(hh, anchor, E, egrad, tgt, V, 1)
    assert abs(loss.item() - ref.item()) <= 2e-3 * abs(ref.item())
    loss.backward()
    scale = hf.grad.abs().max().item()
    close(hh.grad, hf.grad, 3e-2, 2e-2 * scale, "dh")
    close(egrad[:V], Ef.grad, 3e-2, 2e-2 * Ef.grad.abs().max().item(), "dE")
    assert (egrad[V:] == 0).all()


def test_colam(K):
    B, T, D = 6, 16, 1024
    hs = rnd(B, T, D, seed=1); hg = rnd(B, T, D, seed=2)
    hg[0] = hs[0]                               # cos = 1 -> inside the margin only if margin > 1... exercise both sides
    mask = (torch.arange(T)[None, :] < torch.tensor([16, 3, 9, 1, 12, 5])[:, None]).to(torch.uint8).cuda()
    for margin in (1.0, 0.05):
        loss, cos, ps, pg = K.colam_fwd(hs, hg, mask, margin)
        hf = hs.float().requires_grad_(True)
        def pool(h):
            m = mask.bool()
            e = h.masked_fill(~m[..., None], 0.0).sum(1) / m.sum(1)[..., None]
            return torch.nan_to_num(e, nan=1.0)
        a = pool(hf); bb = pool(hg.float())
        a = a / a.norm(dim=1, keepdim=True); bb = bb / bb.norm(dim=1, keepdim=True)
        ref = torch.nn.HingeEmbeddingLoss(margin=margin)((a @ bb.t()).diag(), -torch.ones(B, device="cuda"))
        assert abs(loss.item() - ref.item()) < 1e-4, (loss.item(), ref.item())
        ref.backward()
        g = torch.tensor(1.0, device="cuda")
        dhs = K.colam_bwd(cos, ps, pg, mask, (B, T, D), margin, g, 0.5)
        close(dhs, 0.5 * hf.grad, 2e-2, 1e-6, f"colam bwd margin={margin}")


def test_secla(K):
    B, F, N, D = 8, 4, 6, 1024
    faces = rnd(B, F, D, seed=1); names = rnd(B, N, D, dtype=torch.float32, seed=2)
    loss, sim, l1, l2 = K.secla_fwd(faces, names)
    ff = faces.float().requires_grad_(True)
    def batch_softmax(m):
        bs, _, ns, _ = m.shape
        logits = m.max(-1).values.sum(-1) / ns
        return torch.nn.functional.cross_entropy(logits, torch.arange(bs, device=m.device))
    m1 = torch.matmul(names.unsqueeze(1), ff.permute(0, 2, 1)); m2 = torch.matmul(ff.unsqueeze(1), names.permute(0, 2, 1))
    ref = batch_softmax(m1) + batch_softmax(m2)
    assert abs(loss.item() - ref.item()) < 1e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    ref.backward()
    g = torch.tensor(2.0, device="cuda")
    df = K.secla_bwd(faces, names, sim, l1, l2, g, 1.0)
    close(df, 2.0 * ff.grad, 2e-2, 1e-4, "secla bwd")


# ------------------------------------------------------------------------------------------------- optim
def test_adamw_and_schedule(K):
    n = 4096 + 8
    p = rnd(n, dtype=torch.float32, seed=1); g = rnd(n, dtype=torch.float32, seed=2)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); p16 = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    hyper = torch.zeros(2, device="cuda")
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    total, warm = 100.0, 5.0
    lam = lambda k: k / max(1.0, warm) if k < warm else max(0.0, (total - k) / max(1.0, total - warm))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    for step in range(8):
        gg = g * (1.0 + 0.1 * step)
        ref_p.grad = gg.clone()
        opt.step(); sched.step()
        gbuf = gg.clone()
        K.lr_step(hyper, 3e-5, warm, total)
        K.adamw(p, gbuf, m, v, p16, hyper, n)
        assert (gbuf == 0).all()
    close(p, ref_p.data, 1e-5, 1e-7, "adamw params after 8 steps")
    close(p16, p, 4e-3, 1e-6, "bf16 shadow")
    assert hyper[1].item() == 8.0


@pytest.mark.parametrize("max_norm,world", [(0.1, 1), (1e4, 1), (0.5, 4)])
def test_grad_clip_matches_torch_clip_grad_norm(K, max_norm, world):
    """clip_grad_norm_ + AdamW (TRAIN:365-374) vs the fused device path: coefficient stays on the device, the gradient
    arena is scaled where AdamW reads it.  max_norm=1e4: no clipping (coefficient clamps at 1)."""
    n = 3 * 4096 * 17 + 12
    p = rnd(n, dtype=torch.float32, seed=1); g = rnd(n, dtype=torch.float32, seed=2) * 3.0
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); p16 = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    hyper = torch.zeros(2, device="cuda")
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for step in range(3):
        gg = g * (1.0 + 0.5 * step)                   # the arena holds the SUM over ranks; DDP hands clip_grad_norm_ the mean
        ref_p.grad = gg.clone() / world
        ref_norm = torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
        opt.step()
        gbuf = gg.clone()
        K.lr_step(hyper, 3e-5, 0.0, 1e9)
        out = K.grad_clip_coef(gbuf, n, max_norm, grad_scale=1.0 / world)
        out2 = K.grad_clip_coef(gbuf, n, max_norm, grad_scale=1.0 / world)
        assert torch.equal(out, out2), "the norm must be bit-reproducible"
        assert torch.equal(gbuf, gg), "the gradient arena is not rewritten by the norm pass"
        K.adamw(p, gbuf, m, v, p16, hyper, n, grad_scale=1.0 / world, clip_coef=out)
        assert abs(out[1].item() - ref_norm.item()) <= 1e-5 * ref_norm.item()
        want = min(1.0, max_norm / (ref_norm.item() + 1e-6))
        assert abs(out[0].item() - want) <= 1e-5 * want
        assert (gbuf == 0).all()
    close(p, ref_p.data, 1e-5, 1e-7, "clipped adamw params after 3 steps")


# -------------------------------------------------------------------------------------------------- misc
def test_misc(K):
    ids = torch.tensor([[0, 5, 6, 2, 1], [0, 9, 2, 1, 1]], device="cuda")
    mask, sh = K.prep_ids(ids, pad_id=1, start_id=2)
    assert mask.tolist() == [[1, 1, 1, 1, 0], [1, 1, 1, 0, 0]]
    assert sh.tolist() == [[2, 0, 5, 6, 2], [2, 0, 9, 2, 1]]
    a = rnd(2, 5, 64, seed=1); b = rnd(2, 3, 64, seed=2)
    assert torch.equal(K.cat_tokens([a, b]), torch.cat([a, b], 1))
    assert torch.equal(K.add(a, a), (a.float() * 2).bfloat16())
    lg = rnd(7, 1000, seed=3); lg[2, 10] = 50; lg[2, 900] = 50
    am = K.argmax_rows(lg, 1000)
    assert torch.equal(am, lg.float().argmax(-1)) and am[2].item() == 10
    dy = rnd(300, 520, seed=4); dbias = torch.ones(520, device="cuda")
    K.bias_grad(dy, dbias, 300, 520)
    close(dbias, 1.0 + dy.float().sum(0), 1e-3, 1e-2, "bias grad")
    img = rnd(2, 3, 28, 28, dtype=torch.float32, seed=5)
    pt = K.im2col_patches(img, 14, 592)
    ref = torch.nn.functional.unfold(img, 14, stride=14).transpose(1, 2).reshape(2 * 4, 588)
    close(pt[:, :588], ref, 1e-2, 1e-2, "im2col"); assert (pt[:, 588:] == 0).all()
    face = torch.ones(2, 3, 512, device="cuda"); face[0, 0] = 0.3
    assert K.face_mask(face).tolist() == [[1, 0, 0], [0, 0, 0]]
    x32 = rnd(1000, dtype=torch.float32, seed=6)
    assert torch.equal(K.cast_f32_bf16(x32), x32.bfloat16())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K_", [(5, 1024, 1024), (1, 3072, 1024), (8, 1024, 4096), (5, 50267, 1024), (3, 1000, 520)])
def test_gemm_skinny_rows(K, M, N, K_):
    """single-token decode shapes: M <= 8 goes through the W-streaming kernel (bias, GELU, bf16 and fp32 outputs, ldo > N)."""
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = (torch.randn(M, K_, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K_, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    ref = x.float() @ w.float().t() + b
    tol = 2e-2 * ref.abs().max().item()
    close(K.gemm(x, w, M, N, K_, bias=b), ref, 1e-2, tol, "skinny bf16")
    close(K.gemm(x, w, M, N, K_, bias=b, tile_hint=64), ref, 1e-2, tol, "tile path bf16")
    close(K.gemm(x, w, M, N, K_, bias=b, act="gelu"), torch.nn.functional.gelu(ref), 1e-2, tol, "skinny gelu")
    ldo = (N + 15) // 8 * 8
    out = torch.zeros(M, ldo, device="cuda", dtype=torch.float32)
    K.gemm(x, w, M, N, K_, bias=b, out=out, ldo=ldo, out_mode=1)
    close(out[:, :N], ref, 1e-3, 2e-3 * ref.abs().max().item(), "skinny fp32 out")
    assert out[:, N:].abs().max().item() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K_,act,f32", [(5, 3072, 1024, None, False), (5, 1024, 1024, None, False), (8, 4096, 1024, "gelu", False),
                                             (1, 50267, 1024, None, True), (3, 1000, 768, None, False), (5, 520, 512, "tanh", False)])
def test_gemv_ln_prologue_is_bit_identical_to_add_ln_then_skinny_gemm(K, M, N, K_, act, f32):
    """single-token decoder: residual add + LayerNorm fused as a prologue into the consuming projection."""
    x = rnd(M, K_, seed=1); res = rnd(M, K_, seed=2)
    g = rnd(K_, dtype=torch.float32, seed=3) * 0.5 + 1.0; b = rnd(K_, dtype=torch.float32, seed=4) * 0.1
    w = rnd(N, K_, scale=0.05, seed=5); bias = rnd(N, dtype=torch.float32, seed=6)
    ld = (N + 31) // 32 * 32
    h_ref, _, _ = K.add_ln_fwd(x.view(M, 1, K_), res.view(M, 1, K_), g, b, need_stats=False)
    want = torch.zeros(M, ld, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    K.gemm(h_ref.view(M, K_), w, M, N, K_, bias=bias, out=want, ldo=ld, act=act, out_mode=1 if f32 else 0, tile_hint=8)
    got = torch.zeros_like(want)
    hn = torch.empty(M, K_, device="cuda", dtype=torch.bfloat16)
    K.gemv_ln(x, res, g, b, w, M, N, K_, bias=bias, out=got, ln_out=hn, ldo=ld, act=act, out_mode=1 if f32 else 0)
    assert torch.equal(hn, h_ref.view(M, K_)), "normalised rows"
    assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()
    ref = torch.nn.functional.layer_norm(x.float() + res.float(), (K_,), g, b) @ w.float().t() + bias
    f = {None: lambda t: t, "gelu": torch.nn.functional.gelu, "tanh": torch.tanh}[act]
    close(got[:, :N], f(ref), 2e-2, 2e-2, "vs fp32")
