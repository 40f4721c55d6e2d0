"""Host-side enqueue time of a training step vs. the step time: the host blocks only in the pillar-count readback wait
(PillarSegments.sync_counts); everything else of train_step is Python + launch calls.  host busy = wall - blocked."""
import sys
import time

import torch

sys.path.insert(0, '/root/repo')
import geomae_amd
from geomae_amd import ops, synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer

dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
B = 4
pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
step = lambda i: tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
blocked = [0.0]
orig = ops.PillarSegments.sync_counts


def timed_sync(self):
    t = time.perf_counter()
    r = orig(self)
    blocked[0] += time.perf_counter() - t
    return r


ops.PillarSegments.sync_counts = timed_sync
for i in range(8):
    step(i)
torch.cuda.synchronize()
blocked[0] = 0.0
K = 60
t0 = time.perf_counter()
for i in range(K):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"step {1e3 * (t2 - t0) / K:.3f} ms; host loop {1e3 * (t1 - t0) / K:.3f} ms/step of which blocked in the count readback "
      f"{1e3 * blocked[0] / K:.3f} ms -> host busy {1e3 * (t1 - t0 - blocked[0]) / K:.3f} ms/step")
