"""Host-side timing of the tower graph replays inside a running training loop (does hipGraphLaunch block the host?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import synthetic, streams, training
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
marks = []
for name in ("g_guide", "g_vit"):
    g = getattr(towers, name)
    orig = g.replay
    def mk(orig, name):
        def timed():
            t = time.perf_counter(); orig(); marks.append((name, (time.perf_counter() - t) * 1e3))
        return timed
    class W:            # CUDAGraph.replay is read-only: wrap the object
        def __init__(self, g, f): self.g, self.replay = g, f
    setattr(towers, name, W(g, mk(orig, name)))
orig_launch = towers.launch
def launch(batch, ready):
    t = time.perf_counter(); r = orig_launch(batch, ready); marks.append(("launch total", (time.perf_counter() - t) * 1e3)); return r
towers.launch = launch
orig_fl = training.forward_losses
for i in range(6):
    t = time.perf_counter()
    train_step(model, guide, opt, batches[i % 2], args, ready, towers)
    marks.append(("train_step host", (time.perf_counter() - t) * 1e3))
torch.cuda.synchronize()
for m in marks[-12:]: print(f"{m[0]:18s} {m[1]:8.2f} ms")
