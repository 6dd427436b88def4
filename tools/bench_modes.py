import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import streams, synthetic
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.training import FusedAdamW, GraphedTrainStep, TrainArgs, build_models, to_device, train_step
cfg, vcfg = bart_large_vit_l14()
model, guide, _ = build_models(cfg, vcfg, device="cuda", seed=1, init="device")
args = TrainArgs()
opt = FusedAdamW(model.arena, lr=3e-5, num_warmup_steps=100, num_training_steps=10000)
bs = [to_device(synthetic.make_batch(cfg, 32, S=512, T=64, seed=42, step=i, full_length=True), "cuda") for i in range(2)]
def run(fn, n=6):
    fn(bs[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(bs[i % 2])
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, h / n * 1e3
for st in (False, True):
    streams.enable(st)
    for _ in range(2): train_step(model, guide, opt, bs[0], args)
    print(f"streams={st} eager : %.1f ms/step (host %.1f)" % run(lambda b: train_step(model, guide, opt, b, args)), flush=True)
    g = GraphedTrainStep(model, guide, opt, args, bs[0], warmup=1)
    print(f"streams={st} graph : %.1f ms/step (host %.1f)" % run(g), flush=True)
    del g
