import time, torch, sys
sys.path.insert(0, '/root/repo')
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
B = 4
pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
def step(i): return tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
def bench(tag, K=40):
    for i in range(6): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): l, _ = step(i)
    torch.cuda.synchronize(); print(f"{tag}: {1e3*(time.perf_counter()-t0)/K:.3f} ms/step loss {float(sum(l.values())):.4f}")
for mode in (False, True, False, True):
    type(model).OVERLAP_GEOMETRY = mode
    bench(f"overlap_geometry={mode}")
