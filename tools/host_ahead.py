"""How far ahead of the GPU does the Python launch path run in the bench loop?  Before enqueuing step i, poll the completion
events of steps i-1 and i-2: "i-1 still running" means the host is at least one step ahead (the next step's frozen towers can
then start during the current backward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import streams, synthetic
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.training import FrozenTowerGraphs, FusedAdamW, TrainArgs, build_models, to_device, train_step

cfg, vcfg = bart_large_vit_l14()
streams.enable(True)
model, guide, _ = build_models(cfg, vcfg, seed=1, init="device")
args = TrainArgs(num_training_steps=100000)
opt = FusedAdamW(model.arena, lr=3e-5, num_warmup_steps=5000, num_training_steps=100000)
batches = [to_device(synthetic.make_batch(cfg, 32, S=512, T=64, seed=42, step=i, full_length=True), "cuda") for i in range(4)]
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
towers = FrozenTowerGraphs(model, guide, batches[0])
for i in range(3):
    train_step(model, guide, opt, batches[i % 4], args, ready, towers)
torch.cuda.synchronize()
N = 40
done, host_ms, ahead1, ahead2 = [], [], 0, 0
t0 = time.perf_counter()
for i in range(N):
    if i >= 1 and not done[i - 1].query():
        ahead1 += 1
    if i >= 2 and not done[i - 2].query():
        ahead2 += 1
    h0 = time.perf_counter()
    train_step(model, guide, opt, batches[i % 4], args, ready, towers)
    host_ms.append((time.perf_counter() - h0) * 1e3)
    e = torch.cuda.Event(); e.record(); done.append(e)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
host_ms.sort()
print(f"step {dt:.2f} ms; host enqueue median {host_ms[N // 2]:.1f} ms (min {host_ms[0]:.1f}, max {host_ms[-1]:.1f}); "
      f"previous step still running at enqueue: {ahead1}/{N - 1}; two steps back still running: {ahead2}/{N - 2}")
