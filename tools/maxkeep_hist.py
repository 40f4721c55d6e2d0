"""How full is the fullest window of each layout, step by step?  (the one-launch layer kernels last as long as their largest
bundle: 1-3 tiles for windows that kept <= 48 pillars, 4 tiles above)   Usage: python tools/maxkeep_hist.py [steps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(1234)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
trainer = Trainer(geomae_amd.build_model(cfg).to(dev).train())
pool = [[torch.as_tensor(synth.lidar_frame(10_000 + i * 4 + b), device=dev) for b in range(4)] for i in range(4)]
mk = []
for i in range(steps):
    trainer.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
    s = trainer.get_engine().last_sizes()
    mk.append(s["max_window_keep"])
mk = np.array(mk[2:])
print("steps", len(mk), "max kept pillars per layout: mean", mk.mean(0), "min", mk.min(0), "max", mk.max(0))
print("share of layouts with a window above 48 (a 4-tile bundle):", (mk > 48).mean(0), " above 64:", (mk > 64).mean(0))
print("histogram of the fuller layout:", np.bincount(np.minimum(mk.max(1), 80))[30:])
