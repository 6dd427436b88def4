"""Input-pipeline throughput at BASELINE configs[1] shapes (512-token articles, 224x224 uint8 images, B=32): shard read ->
collate -> pinned staging -> H2D on the copy stream -> uint8->fp32 normalise kernel.  Must exceed the training step's rate."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import data, synthetic

n, B = 1024, 32
path = os.path.join(tempfile.mkdtemp(), "bench.vshard")
t0 = time.perf_counter()
with data.ShardWriter(path) as w:
    for s in synthetic.make_samples(n, seed=1, max_article=512, max_caption=64, image_size=224):
        w.add(s)
t_pack = time.perf_counter() - t0
rd = data.ShardReader(path)
ld = data.PrefetchLoader(rd, B, seed=0, depth=4)
for epoch in range(2):
    ld.set_epoch(epoch)
    torch.cuda.synchronize(); t0 = time.perf_counter(); m = 0
    for batch, ev in ld:
        torch.cuda.current_stream().wait_event(ev)
        m += batch["article_ids"].shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"metric": "input pipeline samples/sec (shard -> collate -> pinned -> H2D -> normalise), 1 loader thread", "value": round(m / dt, 1),
                  "unit": "samples/s", "batch": B, "samples": m, "shard_MB": round(os.path.getsize(path) / 1e6, 1), "pack_samples_per_s": round(n / t_pack, 1)}))
