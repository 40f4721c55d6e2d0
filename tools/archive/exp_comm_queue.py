"""The forced one-rank exchange with k streams taken from torch's pool BEFORE the trainer creates its process groups:
shifts which hardware queue the communicators' streams land on.  Prints what the probes saw and the step time; with
GEOMAE_STREAM_PROBE=0 the unguarded picture.  usage: python tools/exp_comm_queue.py [kmax]"""
import os, subprocess, sys
code = r'''
import os, time, torch, sys
sys.path.insert(0, "/root/repo")
import torch.distributed as dist
os.environ["GEOMAE_FORCE_EXCHANGE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
keep = []
for _ in range(int(os.environ.get("BURN", "0"))):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    keep.append(s)
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
torch.manual_seed(1234)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
B = 4
pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
def step(i): return tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
for i in range(6): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(30): step(i)
torch.cuda.synchronize()
print(f"{1e3*(time.perf_counter()-t0)/30:.3f} ms/step", ops.STREAM_PROBE.get(("cuda", 0)))
dist.destroy_process_group()
'''
kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for probe in ("1", "0"):
    for k in range(kmax + 1):
        env = dict(os.environ, BURN=str(k), GEOMAE_STREAM_PROBE=probe)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"probe={probe} burn={k}:", next((l for l in out.stdout.split("\n") if "ms/step" in l), out.stderr.strip().split("\n")[-1][:300]))
