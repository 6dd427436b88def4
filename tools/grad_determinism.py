"""Is one forward/backward bitwise reproducible?  N passes over the same batch with the same weights (dropout off): per-parameter
count of gradient elements that differ from pass 0, the largest relative difference, and the loss vectors.  Small model of the DDP
test by default; --full for configs[1] geometry at batch 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import streams, synthetic
from vacnic_amd.config import ClipVisionConfig, VacnicConfig
from vacnic_amd.training import TrainArgs, build_models, forward_losses


def main(rank=0, world=1, port=0):
    side = "--no-streams" not in sys.argv
    streams.enable(side)
    if world > 1:
        import torch.distributed as dist
        from vacnic_amd import ddp
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = VacnicConfig(d_model=768, encoder_layers=2, decoder_layers=2, encoder_attention_heads=12, decoder_attention_heads=12,
                       encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=128, dropout=0.0)
    vcfg = ClipVisionConfig(width=128, layers=1, patch_size=16, image_size=32, output_dim=64)
    model, guide, _ = build_models(cfg, vcfg, init="synthetic", seed=3)
    model.train()
    args = TrainArgs()
    full = synthetic.make_batch(cfg, 6, S=48, T=12, F=3, seed=11, image_size=32, full_length=True)
    batch = {k: v[3 * rank:3 * rank + 3].contiguous().cuda() for k, v in full.items()}
    net = model
    wrapped = world > 1 and "--share-only" not in sys.argv       # --share-only: two processes on the card, no reducer at all
    if wrapped:
        model = ddp.DistributedDataParallel(net, bucket_bytes=8 << 20, overlap="--no-overlap" not in sys.argv)
    grads, losses, tops = [], [], []
    # --trace: checksums of the operands / results of the first LayerNorm-backward and GEMM calls of every backward pass
    from vacnic_amd import kernels as K
    trace, state = [], {"on": False}
    ck = lambda t: None if t is None else float(t.double().sum().item())
    if "--trace" in sys.argv:
        ln0, gm0 = K.add_ln_bwd, K.gemm

        def ln(dout, x, residual, gamma, mean, rstd, dgamma, dbeta, **kw):
            r = ln0(dout, x, residual, gamma, mean, rstd, dgamma, dbeta, **kw)
            if state["on"] and len(trace[-1]) < 60:
                trace[-1].append(("ln_bwd", tuple(x.shape), ck(dout), ck(x), ck(mean), ck(rstd), ck(r[0])))
            return r

        def gm(x, w, M, N, Kd, **kw):
            r = gm0(x, w, M, N, Kd, **kw)
            if state["on"] and len(trace[-1]) < 60:
                trace[-1].append(("gemm", (M, N, Kd, kw.get("out_mode", 0), int(kw.get("x_kstrided", False)), int(kw.get("w_kstrided", False))), ck(x), ck(w), ck(kw.get("residual")), ck(kw.get("dact_src")), ck(r)))
            return r
        K.add_ln_bwd, K.gemm = ln, gm
    for i in range(6):
        net.arena.grad.zero_()
        total, out4, out = forward_losses(model, guide, batch, args)
        top = {"H": out["decoder_hidden_states"][-1].detach().clone()}
        out["decoder_hidden_states"][-1].register_hook(lambda g, top=top: top.__setitem__("dH", g.detach().clone()))
        if out.get("hidden_states_face") is not None and out["hidden_states_face"].requires_grad:
            out["hidden_states_face"].register_hook(lambda g, top=top: top.__setitem__("dFace", g.detach().clone()))
        tops.append(top)
        trace.append([]); state["on"] = True
        with torch.autograd.set_multithreading_enabled(False):
            total.backward()
        state["on"] = False
        streams.join_all()
        if wrapped:
            if "--no-reduce" not in sys.argv:
                model.reduce_gradients()
            else:
                torch.cuda.synchronize(); model._finish()
        torch.cuda.synchronize()
        grads.append(net.arena.grad.clone()); losses.append(out4.tolist())
    model = net
    if world > 1:
        dist.destroy_process_group()
    if rank != 0:
        return
    print("side streams", side)
    for i in range(1, len(tops)):
        msg = []
        for k in tops[0]:
            a, b = tops[0][k].float(), tops[i][k].float()
            nd = (a != b).nonzero()
            msg.append(f"{k}: {len(nd)} of {a.numel()} differ" + (f" first at {nd[0].tolist()} {a[tuple(nd[0])].item():.6e} vs {b[tuple(nd[0])].item():.6e}" if len(nd) else ""))
        print(f"pass {i} vs 0 (top of backward): " + "; ".join(msg))
    for i, l in enumerate(losses):
        print("pass", i, l)
    if "--trace" in sys.argv:
        for i in range(1, len(trace)):
            for j, (a, b) in enumerate(zip(trace[0], trace[i])):
                if a != b:
                    print(f"pass {i} vs 0: first differing call #{j}:\n    {a}\n    {b}")
                    if j:
                        print(f"    (previous call: {trace[0][j - 1]})")
                    break
            else:
                print(f"pass {i} vs 0: first {len(trace[0])} traced calls identical")
    names = [(n, p) for n, p in model.named_parameters() if id(p) in model.arena.slots]
    for i in range(1, len(grads)):
        bad = []
        for n, p in names:
            o, cnt, _ = model.arena.slots[id(p)]
            a, b = grads[0][o:o + cnt], grads[i][o:o + cnt]
            nd = int((a != b).sum().item())
            if nd:
                bad.append((((a - b).norm() / a.norm().clamp_min(1e-30)).item(), nd, cnt, n))
        print(f"pass {i} vs 0: {len(bad)} of {len(names)} tensors differ")
        for r in sorted(bad, reverse=True)[:(200 if "--all" in sys.argv else 12)]:
            print(f"    rel {r[0]:.2e}  {r[1]}/{r[2]}  {r[3]}")


if __name__ == "__main__":
    if "--ddp" in sys.argv:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        port = 29900 + os.getpid() % 1000
        ps = [ctx.Process(target=main, args=(r, 2, port)) for r in range(2)]
        for p_ in ps: p_.start()
        for p_ in ps: p_.join(300)
    else:
        main()
