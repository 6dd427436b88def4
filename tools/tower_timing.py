"""When do the frozen towers of step N+1 actually start on the GPU relative to step N's backward / AdamW, and when did the host
enqueue them?  (events on the tower streams vs events on the compute stream)"""
import sys, os, time
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
E = lambda: torch.cuda.Event(enable_timing=True)
rec = []
orig_launch = towers.launch
def launch(batch, rdy):
    host_t = time.perf_counter()
    vis = streams.vit_stream()
    ev = E(); r = orig_launch(batch, rdy)
    ev2 = E(); ev2.record(vis)          # after the ViT replay on its stream = ViT end
    rec[-1].update(host_launch=host_t, vit_end=ev2)
    return r
towers.launch = launch
t_origin = E(); t_origin.record()
h0 = time.perf_counter()
for i in range(8):
    rec.append({})
    s = E(); s.record()                                   # compute stream position at host entry of the step
    rec[-1]["main_enter"] = s
    train_step(model, guide, opt, batches[i % 2], args, ready, towers)
    e = E(); e.record()
    rec[-1]["main_end"] = e
    rec[-1]["host_end"] = time.perf_counter()
torch.cuda.synchronize()
for i, r in enumerate(rec):
    print(f"step {i}: host launch of towers at {1e3*(r['host_launch']-h0):7.1f} ms, host step end {1e3*(r['host_end']-h0):7.1f} | GPU: main reaches step entry {t_origin.elapsed_time(r['main_enter']):7.1f}, ViT done {t_origin.elapsed_time(r['vit_end']):7.1f}, step end {t_origin.elapsed_time(r['main_end']):7.1f}")
