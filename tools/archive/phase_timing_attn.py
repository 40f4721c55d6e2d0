"""Per-phase cycles inside win_attn_bwd_kernel during a real training step (library built by tools/build_timing.py)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from geomae_amd import _lib
lib = _lib.load(path='/root/repo/tools/libgeomae_timing.so')
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
dev = torch.device('cuda:0')
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(4)]
lib.geomae_debug_read_attn_stamps.argtypes = [ctypes.c_void_p]
def read():
    buf = np.zeros(512 * 16, dtype=np.uint64)
    lib.geomae_debug_read_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    return buf.reshape(512, 16).astype(np.int64)
for i in range(3): tr.train_step(pts)
read(); tr.train_step(pts); st = read()        # last launch of the step = encoder layer 0 backward
ok = st[:, 6] > 0
s = st[ok]
names = ["bundle setup (3 dependent index loads) + barrier", "gather 5 head slices + LDS staging + delta", "pass 1 (dQ)", "store dQ", "(barrier)", "pass 2 (dK, dV)", "store dK, dV"]
print("workgroups stamped", int(ok.sum()))
for k in range(1, 7):
    d = s[:, k] - s[:, k - 1]
    print(f"{names[k-1]:55s} median {np.median(d):8.0f} cycles  mean {d.mean():8.0f}")
print("total median", np.median(s[:, 6] - s[:, 0]))
