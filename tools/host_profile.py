"""cProfile of the host side of train_step (where do the ~50 ms of Python/ctypes enqueue time per step go?)."""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import synthetic, streams
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.training import FrozenTowerGraphs, FusedAdamW, TrainArgs, build_models, to_device, train_step
streams.enable(True)
cfg, vcfg = bart_large_vit_l14()
model, guide, _ = build_models(cfg, vcfg, device="cuda", seed=1234, init="device")
args = TrainArgs(num_training_steps=100000)
opt = FusedAdamW(model.arena, lr=args.lr_bart, weight_decay=args.weight_decay, num_warmup_steps=100, num_training_steps=100000, world_size=1)
batches = [to_device(synthetic.make_batch(cfg, 32, S=512, T=64, seed=42, rank=0, step=i, full_length=True), "cuda") for i in range(2)]
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
towers = FrozenTowerGraphs(model, guide, batches[0])
for i in range(3):
    train_step(model, guide, opt, batches[i % 2], args, ready, towers)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    train_step(model, guide, opt, batches[i % 2], args, ready, towers)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"unprofiled: host {host/3*1e3:.1f} ms/step, wall {(time.perf_counter()-t0)/3*1e3:.1f} ms/step")
if os.environ.get("ST") == "1":
    torch.autograd.set_multithreading_enabled(False)
    t0 = time.perf_counter()
    for i in range(3):
        train_step(model, guide, opt, batches[i % 2], args, ready, towers)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"single-thread autograd: host {host/3*1e3:.1f} ms/step, wall {(time.perf_counter()-t0)/3*1e3:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    train_step(model, guide, opt, batches[i % 2], args, ready, towers)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
