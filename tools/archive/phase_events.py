"""Real (unprofiled) phase times of the explicit training step: HIP events recorded on the main stream at the phase
boundaries (ops.mark), averaged over steps.  The gap between one step's optimizer_done and the next step_start mark is
reported as 'between steps'."""
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
for i in range(8):
    step(i)
torch.cuda.synchronize()
K = 40
ops.PHASE_MARKS = []
t0 = time.perf_counter()
for i in range(K):
    step(i)
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t0) / K
marks, ops.PHASE_MARKS = ops.PHASE_MARKS, None
main = [(n, e) for n, e in marks if not n.startswith("side:")]
sidem = [(n, e) for n, e in marks if n.startswith("side:")]
per = len(main) // K
names = [n for n, _ in main[:per]]
assert all(main[k * per + j][0] == names[j] for k in range(K) for j in range(per)), names
acc = {}
for k in range(K):
    ev = [e for _, e in main[k * per:(k + 1) * per]]
    for j in range(1, per):
        acc[names[j]] = acc.get(names[j], 0.0) + ev[j - 1].elapsed_time(ev[j])
    if k:
        acc["between steps"] = acc.get("between steps", 0.0) + main[k * per - 1][1].elapsed_time(ev[0])
print(f"wall {wall:.3f} ms/step (with {per} event records per step on the main stream)")
for n, v in acc.items():
    print(f"  -> {n:16s} {1e3 * v / (K - 1 if n == 'between steps' else K):8.1f} us")
ps = len(sidem) // K
if ps:
    print("side stream marks, offset from the same step's step_start mark:")
    for j in range(ps):
        tot = sum(main[k * per][1].elapsed_time(sidem[k * ps + j][1]) for k in range(K))
        print(f"  @ {sidem[j][0]:16s} {1e3 * tot / K:8.1f} us")
