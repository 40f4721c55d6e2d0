"""Experiment: the step's main stream at high priority (its kernels dispatched ahead of the side streams' when both have
workgroups ready) versus the default.  One subprocess per setting."""
import os, subprocess, sys
code = r'''
import os, time, torch, sys
sys.path.insert(0, "/root/repo")
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(1234)
prio = os.environ.get("MAIN_PRIO")
main = torch.cuda.Stream(device=dev, priority=int(prio)) if prio is not None else torch.cuda.current_stream(dev)
with torch.cuda.stream(main):
    cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
    model = geomae_amd.build_model(cfg).to(dev).train()
    tr = Trainer(model)
    B = 4
    pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
    torch.cuda.synchronize()
    def step(i): return tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
    for rep in range(2):
        for i in range(6): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): l, _ = step(i)
        torch.cuda.synchronize(); print(f"{1e3*(time.perf_counter()-t0)/40:.3f} ms/step loss {float(sum(l.values())):.4f}")
'''
for val in (None, "-1", None, "-1"):
    env = dict(os.environ)
    if val is not None: env["MAIN_PRIO"] = val
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"MAIN_PRIO={val}:", " | ".join(out.stdout.strip().split("\n")[-2:]), out.stderr.strip().split("\n")[-1][:300] if out.returncode else "")
