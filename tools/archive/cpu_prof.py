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
for i in range(6): step(i)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for i in range(K): step(6 + i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step, drain {1e3*(t2-t1):.3f} ms")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    step(i)
    torch.cuda.synchronize()     # so that waits do not pollute the CPU profile of the next step
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(32)
