"""Phase stamps of heads_loss_kernel inside a training step (timing build: python tools/build_timing.py): per kind of
workgroup (regression chunks 0-2 / class + medium + top chunks 3-5 / the density decoder's normal head) the cycles of
GEMM, loss arithmetic and dX GEMM per chunk."""
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
tr = Trainer(model, optimizer_cfg=dict(type="AdamW", lr=1e-4, weight_decay=0.05))
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pool = [[torch.as_tensor(synth.lidar_frame(10000 + 4 * i + b, sweeps=sweeps), device=dev) for b in range(4)] for i in range(2)]
for i in range(6):
    tr.train_step(pool[i % 2], next_points=pool[(i + 1) % 2])
torch.cuda.synchronize()
lib.geomae_debug_read_heads_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(512 * 32, dtype=np.uint64)
lib.geomae_debug_read_heads_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
tr.train_step(pool[0], next_points=pool[1]); torch.cuda.synchronize()
lib.geomae_debug_read_heads_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
st = buf.reshape(512, 32).astype(np.int64)
for name, sb, end, nch in (("regression chunks 0-2", 0, 30, 3), ("chunks 3-5 (class, medium, top)", 10, 31, 3), ("density head", 20, 29, 1)):
    s = st[(st[:, sb] > 0) & (st[:, end] > 0)]
    if not len(s): continue
    print(f"{name}: {len(s)} workgroups, whole body mean {(s[:, end] - s[:, sb]).mean():.0f} max {(s[:, end] - s[:, sb]).max()} cycles")
    prev = sb
    for k in range(nch):
        a, b, c = sb + 1 + 3 * k, sb + 2 + 3 * k, sb + 3 + 3 * k
        print(f"   chunk {k}: loads + GEMM {(s[:, a] - s[:, prev]).mean():8.0f} | loss arithmetic {(s[:, b] - s[:, a]).mean():8.0f} | dX GEMM {(s[:, c] - s[:, b]).mean():8.0f}")
        prev = c
    print(f"   epilogue (dX rows, loss sums) {(s[:, end] - s[:, prev]).mean():8.0f}")
