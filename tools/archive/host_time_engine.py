"""Host-side enqueue time of a training step through the C engine vs. the step time (tools/host_time.py is the same
measurement of the Python explicit schedule).  Optional argv[1] = number of busy-loop competitor processes pinned to
the SAME core as this process (a crude slow-host model: N competitors ~ a host N+1 times slower)."""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer

hogs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
use_engine = os.environ.get("GEOMAE_NO_ENGINE") != "1"
procs = []
if hogs:
    os.sched_setaffinity(0, {2})
    for _ in range(hogs):
        procs.append(subprocess.Popen(["taskset", "-c", "2", sys.executable, "-c", "while True: pass"]))
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
tr.use_engine = use_engine
B = 4
pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
step = lambda i: tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
try:
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    h0 = tr.engine.host_times() if tr.engine else (0, 0, 0)
    K = 60
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    h1 = tr.engine.host_times() if tr.engine else (0, 0, 0)
    msg = f"engine={use_engine} hogs={hogs}: step {1e3 * (t2 - t0) / K:.3f} ms; host loop {1e3 * (t1 - t0) / K:.3f} ms/step"
    if tr.engine:
        msg += (f"; inside geomae_pretrain_step {1e3 * (h1[0] - h0[0]) / K:.3f} ms/step of which blocked on the count readback "
                f"{1e3 * (h1[1] - h0[1]) / K:.3f} -> host busy {1e3 * ((t1 - t0) - (h1[1] - h0[1])) / K:.3f} ms/step")
    print(msg)
finally:
    for p in procs:
        p.kill()
