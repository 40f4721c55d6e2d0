"""Phase cycles inside sst_layer_bwd_kernel (timing build: python tools/build_timing.py) from clock64 stamps of wave 0 and
wave 4 of every workgroup of the LAST backward layer of one engine step.  Usage: python tools/fused_bwd_time.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
lib = _lib.load(path=os.path.join(ROOT, "tools", "libgeomae_timing.so"))
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
tr = Trainer(model)
pool = [[torch.as_tensor(synth.lidar_frame(10000 + 4 * i + b), device=dev) for b in range(4)] for i in range(3)]
for i in range(4):
    tr.train_step(pool[i % 3], next_points=pool[(i + 1) % 3])
torch.cuda.synchronize()
NAMES = ["plan, dz / xhat2 / rstd, LN2 sums", "barrier 1", "LN2 bwd, dv rows, hp / xhat1 loads", "barrier 2", "dh = d W2, gelu', stores",
         "barrier 3", "dy = d + dhp W1, LN1 sums, qkv loads", "barrier 4", "LN1 bwd, du rows", "barrier 5", "dO = du Wo, head slices",
         "attention pass 1 (dQ)", "attention pass 2 (dK, dV)", "barriers 6, 7 + dqkv rows", "dx = du + dqkv Wqkv, store"]
buf = np.zeros(512 * 32, dtype=np.uint64)
lib.geomae_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0)
st = buf.reshape(512, 32).astype(np.int64)
for half in (0, 1):
    s = st[:, 16 * half:16 * half + 16]
    s = s[(s[:, 0] > 0) & (s[:, 15] > s[:, 0])]
    tot = s[:, 15] - s[:, 0]
    print(f"wave {4 * half}: {len(s)} workgroups, total per workgroup mean {tot.mean():.0f} median {np.median(tot):.0f} max {tot.max()} cycles")
    for k, nm in enumerate(NAMES):
        d = s[:, k + 1] - s[:, k]
        print(f"    {nm:44s} mean {d.mean():8.0f}  med {np.median(d):8.0f}  max {d.max():8.0f}")
