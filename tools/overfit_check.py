"""Sanity run of the whole training loop: 120 steps on ONE fixed batch at a raised learning rate; the six losses must fall
(forward, backward, clip and AdamW all have to be right for that)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model, optimizer_cfg=dict(type="AdamW", lr=2e-4, weight_decay=0.05))
pts = [torch.as_tensor(synth.lidar_frame(500 + b), device=dev) for b in range(4)]
for i in range(121):
    l, g = tr.train_step(pts, next_points=pts)
    if i % 20 == 0:
        print(i, round(float(sum(l.values())), 4), {k: round(float(v), 3) for k, v in l.items()}, 'gnorm', round(float(g), 3), flush=True)
