"""Long run of the C step engine: 3000 steps over 7 cycling batches of different sizes (incl. one that forces a workspace
re-creation and one B-changing detour), checking that losses stay finite and fall, that the allocator footprint is flat
and that the per-step time has no drift or stalls."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model, optimizer_cfg=dict(type="AdamW", lr=1e-4, weight_decay=0.05))
pool = [[torch.as_tensor(synth.lidar_frame(900 + 10 * i + b, n_az=900 + 40 * i), device=dev) for b in range(4)] for i in range(7)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
times, first, last = [], None, None
mem0 = None
t_prev = time.perf_counter()
for i in range(steps):
    losses, gnorm = tr.train_step(pool[i % 7], next_points=pool[(i + 1) % 7])
    if i % 100 == 99:
        tot = float(sum(v for v in losses.values()))          # syncs
        now = time.perf_counter()
        times.append((now - t_prev) / 100 * 1e3)
        t_prev = now
        assert np.isfinite(tot) and np.isfinite(float(gnorm)), (i, tot)
        first = tot if first is None else first
        last = tot
        if i == 199:
            mem0 = torch.cuda.memory_reserved(dev)
        if i > 199:
            assert torch.cuda.memory_reserved(dev) == mem0, "allocator footprint moved"
print(f"{steps} steps: loss {first:.3f} -> {last:.3f}; ms/step per 100-step window: min {min(times):.3f} median {np.median(times):.3f} "
      f"max {max(times):.3f}; reserved {mem0 / 2**20:.0f} MiB; optimizer steps {tr.engine.last_sizes()['optimizer_steps']}")
assert last < first and max(times[1:]) < 1.1 * np.median(times)      # (the first window holds the engine's creation)
