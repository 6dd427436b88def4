"""bench.py's config-5 id check alone (full-size model, random init): python tools/cfg5_idcheck.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.training import build_models
cfg, vcfg = bart_large_vit_l14()
model, _, _ = build_models(cfg, vcfg, device="cuda", seed=1234, init="device", with_guide=False)
model.eval()
print(json.dumps(bench.config5_id_check(model, cfg), indent=1))
