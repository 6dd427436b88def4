"""Which ATen operators does one training step still dispatch?  (A launch plan records C-ABI calls only: an ATen kernel inside the
step would be missing from a replay.)  python tools/aten_in_step.py [--no-streams]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from vacnic_amd import streams, synthetic
from vacnic_amd.config import ClipVisionConfig, VacnicConfig
from vacnic_amd.training import FusedAdamW, TrainArgs, build_models, to_device, train_step

META = ("view", "reshape", "empty", "as_strided", "detach", "alias", "slice", "select", "unsqueeze", "squeeze", "transpose", "expand",
        "_unsafe_view", "t.default", "permute", "unbind", "split", "is_", "sym_", "size", "stride", "_local_scalar", "lift_fresh", "unflatten")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.examples = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(m in name for m in META):
            self.ops[name] += 1
            if name not in self.examples:
                import traceback
                fr = [f for f in traceback.extract_stack() if "/vacnic_amd/" in f.filename or "/tools/" in f.filename]
                self.examples[name] = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-4:])
        return func(*args, **(kwargs or {}))


streams.enable("--no-streams" not in sys.argv)
cfg = VacnicConfig(d_model=768, encoder_layers=2, decoder_layers=2, encoder_attention_heads=12, decoder_attention_heads=12, encoder_ffn_dim=3072,
                   decoder_ffn_dim=3072, enc_fusion_layer=[0, 1], dim_common=768, clip_width=768, dropout=0.1)
vcfg = ClipVisionConfig(width=768, layers=1, patch_size=16, image_size=32, output_dim=64)
args = TrainArgs(num_training_steps=20)
model, guide, _ = build_models(cfg, vcfg, init="synthetic", seed=0)
opt = FusedAdamW(model.arena, lr=1e-4, num_warmup_steps=2, num_training_steps=20)
b = to_device(synthetic.make_batch(cfg, 3, S=32, T=12, F=3, seed=40, image_size=32), "cuda")
for _ in range(2):
    train_step(model, guide, opt, b, args)
torch.cuda.synchronize()
with Log() as log:
    train_step(model, guide, opt, b, args)
torch.cuda.synchronize()
for k, v in log.ops.most_common():
    print(f"{v:5d}  {k:45s} {log.examples[k]}")
