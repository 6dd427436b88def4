"""bench.py — VACNIC train-step throughput on MI355X (BASELINE.json metric: train samples/sec,
BART-large + CLIP ViT-L/14, GoodNews-shaped batch; config.workload names configs[1]/[2]).

    python bench.py --gpus 1 --steps 8 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full training step on one synthetic batch already resident in HBM: ViT-L/14 features,
multimodal BART-large forward (+ fused LM-head/CE), frozen guide BART forward, CoLaM + SECLA losses,
backward, gradient all-reduce (N > 1), fused AdamW + LR schedule.  Nothing is skipped or cached.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md
STEP_GFLOP_PER_SAMPLE = {512: 1179.8, 1024: 2142.7}      # SURVEY §8d (3*F_t + F_g + F_v)


T0 = time.time()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE: 32)")
    ap.add_argument("--seq", type=int, default=512, help="article tokens S")
    ap.add_argument("--cap", type=int, default=64, help="caption tokens T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--grad-transport", choices=("fp32", "bf16"), default="fp32",
                    help="N>1: dtype of the gradient all-reduce (fp32 = the reference's DDP; bf16 halves the xGMI volume)")
    ap.add_argument("--no-streams", action="store_true", help="single-stream schedule (no side streams for guide / wgrad)")
    ap.add_argument("--no-tower-graphs", action="store_true", help="launch the frozen guide/ViT forwards eagerly instead of as two hipGraph replays")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (world 1 only). Measured slower than "
                    "eager multi-stream launches while the step is GPU-bound (91.0 vs 86.2 ms: hipGraph runs the side-stream branches "
                    "less concurrently), so eager is the default")
    return ap.parse_args()


class GemmTimer:
    """HIP events around every GEMM launch of ONE timed step, on the stream the kernels are launched on."""

    def __init__(self):
        self.rec = []

    def install(self):
        from vacnic_amd import kernels as K
        self.orig = K.gemm
        timer = self

        def timed(x, w, M, N, Kd, **kw):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            out = timer.orig(x, w, M, N, Kd, **kw)
            e.record()
            kind = ("T" if kw.get("x_kstrided") else "N") + ("T" if kw.get("w_kstrided") else "N")
            timer.rec.append((kind, 2.0 * M * N * Kd, s, e, (M, N, Kd)))
            return out
        K.gemm = timed

    def remove(self):
        from vacnic_amd import kernels as K
        K.gemm = self.orig

    def summary(self):
        agg = {}
        shapes = {}
        for kind, fl, s, e, shp in self.rec:
            dt = s.elapsed_time(e) * 1e-3
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            a[0] += fl; a[1] += dt; a[2] += 1
            b = shapes.setdefault((kind,) + shp, [0.0, 0.0, 0])
            b[0] += fl; b[1] += dt; b[2] += 1
        if os.environ.get("VACNIC_BENCH_SHAPES"):
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:40]:
                log(f"  gemm {k[0]} M={k[1]:6d} N={k[2]:6d} K={k[3]:6d} x{v[2]:3d}: {v[1]*1e3:7.2f} ms  {v[0]/v[1]/1e12:7.1f} TF/s")
        return agg


def cpu_baseline(cfg, vcfg, B, S, T):
    """The oracle (CPU restatement, torch fp32 + autograd) timed on this box's host cores: fwd + bwd + AdamW of the
    same step at the same shapes, reduced batch.  Reported beside the GPU number, never the target."""
    from oracle import vacnic_oracle as O
    from vacnic_amd import synthetic
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)                      # the box's CPU share for one GPU; more threads only thrash
    torch.set_num_threads(cores)
    pool = torch.randn(1 << 24, generator=torch.Generator().manual_seed(0)) * 0.02

    def rand_sd(shapes):
        # timing only: weights are slices/tiles of one random pool (drawing 1.6 G normals would take a minute)
        out = {}
        for k, v in shapes.items():
            n = 1
            for d in v:
                n *= d
            if len(v) > 1:
                out[k] = (pool[:n] if n <= pool.numel() else pool.repeat((n + pool.numel() - 1) // pool.numel())[:n]).clone().view(v)
            else:
                out[k] = torch.ones(v) if k.endswith("weight") else torch.zeros(v)
        return out
    sd = rand_sd(synthetic.mmbart_param_shapes(cfg))
    sd_g = rand_sd(synthetic.guide_bart_param_shapes(cfg))
    sd_c = rand_sd(synthetic.clip_visual_param_shapes(vcfg))
    for v in sd.values():
        v.requires_grad_(True)
    batch = synthetic.make_batch(cfg, B, S=S, T=T, seed=1, full_length=True)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    vv = {k: torch.zeros_like(v) for k, v in sd.items()}

    def step(i):
        res = O.train_losses(sd, sd_g, sd_c, cfg, vcfg, batch)
        res["loss"].backward()
        with torch.no_grad():
            for k, p in sd.items():
                if p.grad is None:
                    continue
                np_, m[k], vv[k] = O.adamw_step(p, p.grad, m[k], vv[k], i + 1, 3e-5)
                p.copy_(np_); p.grad = None
    log("  oracle weights ready; warm-up step")
    step(0)                                     # warm-up
    best = 1e30
    for i in range(1):
        t0 = time.time(); step(i + 1); best = min(best, time.time() - t0)
        log(f"  oracle step {i}: {time.time() - t0:.1f}s")
    return {"value": round(B / best, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch fp32 CPU, {cores} threads) full train step fwd+bwd+AdamW, same model/shapes, batch {B} "
                      f"(S={S}, T={T}), 1 timed step after 1 warm-up"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")
    # VACNIC_DIST_BACKEND=gloo VACNIC_SINGLE_DEVICE=1: rehearse the N>1 code path (tracker, bucket launches, side streams,
    # joins) with several ranks sharing ONE GPU — RCCL itself needs one GPU per rank, which only the driver's node has
    single_dev = os.environ.get("VACNIC_SINGLE_DEVICE") == "1"
    backend = os.environ.get("VACNIC_DIST_BACKEND", "nccl")
    torch.cuda.set_device(0 if single_dev else local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from vacnic_amd import synthetic
    from vacnic_amd.config import bart_large_vit_l14
    from vacnic_amd.ddp import DistributedDataParallel
    from vacnic_amd.training import FrozenTowerGraphs, FusedAdamW, GraphedTrainStep, TrainArgs, build_models, to_device, train_step

    from vacnic_amd import streams
    streams.enable(not a.no_streams)
    cfg, vcfg = bart_large_vit_l14()
    B, S, T = a.batch, a.seq, a.cap
    log("building models (random init on device)")
    model, guide, _ = build_models(cfg, vcfg, device="cuda", seed=1234, init="device")
    log(f"models ready: {model.arena.n/1e6:.1f}M trainable, {guide.arena.n/1e6:.1f}M guide, {model.clip_model.visual.arena.n/1e6:.1f}M ViT")
    args = TrainArgs(num_training_steps=100000)
    opt = FusedAdamW(model.arena, lr=args.lr_bart, weight_decay=args.weight_decay,
                     num_warmup_steps=args.warmup_rate * args.num_training_steps, num_training_steps=args.num_training_steps,
                     world_size=world)
    nb = 4
    batches = [to_device(synthetic.make_batch(cfg, B, S=S, T=T, seed=42, rank=rank, step=i, full_length=True), "cuda") for i in range(nb)]
    torch.cuda.synchronize()
    ready = torch.cuda.Event()
    ready.record()                      # the synthetic batches are resident and complete: frozen towers may start on this

    towers = None
    if not a.no_streams and not a.no_tower_graphs and not a.graph:
        try:
            towers = FrozenTowerGraphs(model, guide, batches[0])
            log("frozen towers (guide BART, CLIP ViT) captured as hipGraphs")
        except Exception as e:
            log(f"tower graph capture failed ({e!r}); eager launches")
            towers = None
    # the process group comes up only now: the tower graphs above are captured with no communicator (and no watchdog thread
    # issuing HIP calls) alive in the process
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    net = DistributedDataParallel(model, grad_transport=a.grad_transport) if world > 1 else model
    use_graph = world == 1 and a.graph
    log(f"batches resident; warm-up ({'hipGraph capture' if use_graph else 'eager'})")
    graphed = None
    if use_graph:
        try:
            graphed = GraphedTrainStep(net, guide, opt, args, batches[0], warmup=max(1, a.warmup - 1))
            graphed(batches[1 % nb])
            torch.cuda.synchronize()
            log("graph captured and replayed once")
        except Exception as e:                      # never lose the measurement to a capture problem
            log(f"graph capture failed ({e!r}); falling back to eager launches")
            graphed = None
            torch.cuda.synchronize()
    if graphed is None:
        for i in range(a.warmup):
            train_step(net, guide, opt, batches[i % nb], args, ready, towers)
            torch.cuda.synchronize()
            log(f"warm-up step {i} done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer = GemmTimer()
    t0 = time.perf_counter()
    out4 = None
    for i in range(a.steps):
        bt = batches[(a.warmup + i) % nb]
        if i == a.steps - 1:
            # last timed step: single-stream eager launches with HIP events around every GEMM launch, so each bracket times
            # ONE kernel alone on the GPU (with side streams the brackets would overlap other kernels) — this is the
            # roofline measurement of the dominant kernel, taken inside the timed region (it costs ~12 ms of throughput once)
            streams.enable(False)
            timer.install()
            out4 = train_step(net, guide, opt, bt, args, ready, None)
            streams.enable(not a.no_streams)
        elif graphed is not None:
            out4 = graphed(bt)
        else:
            out4 = train_step(net, guide, opt, bt, args, ready, towers)
    timer.remove()
    host_dt = time.perf_counter() - t0           # host-side enqueue time of the K steps (GPU may still be running)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    losses = out4.tolist()
    log(f"timed {a.steps} steps in {dt:.3f}s (host enqueue {host_dt / a.steps * 1e3:.1f} ms/step); losses {losses}")
    if rank == 0:
        agg = timer.summary()
        dom = max(agg.items(), key=lambda kv: kv[1][1]) if agg else None
        roof = None
        if dom:
            kind, (fl, sec, n) = dom
            ach = fl / sec / 1e12
            names = {"NN": "gemm_kernel<false,false> (forward Linear, X[M,K] W[N,K])", "NT": "gemm_kernel<false,true> (dgrad)",
                     "TT": "gemm_kernel<true,true> (wgrad)", "TN": "gemm_kernel<true,false>"}
            traffic = None
            try:        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes of this same command
                pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.json")))["kernels"]
                tag = {"NN": "false, false>", "NT": "false, true>", "TT": "true, true>", "TN": "true, false>"}[kind]
                sel = [v for k, v in pm.items() if "gemm_kernel" in k and tag in k]
                if sel:
                    traffic = round(sum(v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in sel) / sum(v["launches"] for v in sel))
            except Exception:
                traffic = None
            roof = {"bound": "mfma", "kernel": names[kind], "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_pmc_hbm_traffic.json)", "launches": n,
                    "flop_per_launch": round(fl / n),
                    "avg_launch_us": round(sec / n * 1e6, 2),
                    "all_gemm": {k: {"TFLOP/s": round(v[0] / v[1] / 1e12, 1), "ms": round(v[1] * 1e3, 2), "launches": v[2]} for k, v in agg.items()}}
        samples = B * world * a.steps
        value = samples / dt
        gf = STEP_GFLOP_PER_SAMPLE.get(S)
        res = {"metric": "train samples/sec, BART-large + CLIP ViT-L/14 full VACNIC step, GoodNews-shaped batch",
               "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"BASELINE configs[{1 if world == 1 else 2}]: BART-large + CLIP ViT-L/14 full VACNIC (clipcap P=20, SECLA, CoLaM a=0.5 m=1.0), "
                                      f"224x224 image, {S}-token article, {T}-token caption, per-GPU batch {B}, dropout 0.1, fp32 master + bf16 compute",
                          "global_batch": B * world, "seq_len": S, "caption_len": T, "parallelism": f"dp{world}"},
               "launch_mode": ("hipGraph replay (K-1 steps) + 1 eager instrumented step" if graphed is not None else
                               "eager multi-stream" + (", frozen towers as hipGraph replays" if towers is not None else "")),
               "host_enqueue_ms_per_step": round(host_dt / a.steps * 1e3, 2),
               "step_tflops_per_gpu": round(value / world * gf / 1e3, 1) if gf else None,
               "step_mfma_frac": round(value / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4) if gf else None,
               "losses_last_step": {"total": losses[0], "txt": losses[1], "secla": losses[2], "colam": losses[3]},
               "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            try:
                log("cpu baseline (oracle on host cores)")
                res["cpu_baseline"] = cpu_baseline(cfg, vcfg, a.cpu_batch, S, T)
                log("cpu baseline done")
            except Exception as e:      # the baseline must never sink the GPU measurement
                res["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
