"""Host time of the pieces of the backward pass (they run in the autograd engine's thread, invisible to cProfile)."""
import time, torch, sys, collections
sys.path.insert(0, '/root/repo')
import geomae_amd
from geomae_amd import synth, ops, sst
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * 4 + b), device=dev) for b in range(4)] for i in range(4)]
acc = collections.Counter(); cnt = collections.Counter()
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
    setattr(mod, name, g)
for name in ("vfe_backward", "sst_stack_backward", "heads_weight_grad", "vfe_forward", "sst_stack_forward", "heads_loss",
             "window_build", "geometry_targets", "random_mask", "pillar_segment", "voxelize_batch3", "pack_weights"):
    wrap(ops, name)
def step(i): return tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
for i in range(6): step(i)
torch.cuda.synchronize(); acc.clear(); cnt.clear()
K = 20
t0 = time.perf_counter()
for i in range(K):
    tb = time.perf_counter()
    step(i)
    torch.cuda.synchronize()
acc_total = time.perf_counter() - t0
for k, v in acc.most_common(): print(f"{k:22s} {1e3*v/K:7.3f} ms/step  ({cnt[k]//K} calls)")
