"""400 training steps over 6 cycling batches: losses stay finite and fall, allocator footprint stays flat."""
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
tr = Trainer(model)
pool = [[torch.as_tensor(synth.lidar_frame(900 + 4 * i + b), device=dev) for b in range(4)] for i in range(6)]
for i in range(400):
    l, g = tr.train_step(pool[i % 6], next_points=pool[(i + 1) % 6])
    if i % 100 == 0 or i == 399:
        torch.cuda.synchronize()
        tot = float(sum(l.values()))
        assert tot == tot and abs(tot) < 1e6, tot
        print(i, round(tot, 4), 'alloc MB', torch.cuda.memory_allocated() >> 20, 'reserved MB', torch.cuda.memory_reserved() >> 20, flush=True)
