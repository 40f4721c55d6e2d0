import ctypes, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from geomae_amd import _lib
lib = _lib.load(path='/root/repo/tools/libgeomae_timing.so')
from geomae_amd import ops
import geomae_amd
from geomae_amd.configs import mae_sst_model
dev = torch.device('cuda:0')
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
lib.geomae_debug_read_stamps.restype = ctypes.c_int
lib.geomae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
SL, NB = 32, 512
def read():
    buf = np.zeros(NB * SL, dtype=np.uint64)
    lib.geomae_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(NB, SL)
from geomae_amd import synth
from geomae_amd.train import Trainer
tr = Trainer(model)
B = 4
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(B)]
for i in range(3): tr.train_step(pts)
read()
tr.train_step(pts)
st = read()   # stamps of the LAST launch of each kernel (= first encoder layer's backward for ffn_bwd)
names = {0: 'start', 1: 'ld dz/xh2 + LN2bwd + store dv', 2: 'w2T: loads issued', 3: 'w2T: barrier1 (prev consumed)', 4: 'w2T: LDS written+barrier2',
         5: 'w2T mfma done', 6: 'hp load + gelu + stores', 7: 'w1T loads issued', 8: 'w1T barrier1', 9: 'w1T lds+barrier2', 10: 'w1T mfma done',
         11: 'LN1 bwd + stores', 12: 'woT loads issued', 13: 'woT barrier1', 14: 'woT lds+barrier2', 15: 'woT mfma', 16: 'store dattn', 17: 'red flush'}
nblk = int((st[:, 0] > 0).sum())
print("blocks stamped", nblk)
s = st[:nblk].astype(np.int64)
t0 = s[:, 0:1]
rel = s - t0
prev = 0
order = sorted(names)
print("phase deltas (cycles of s_memtime, mean over blocks; median)")
last = np.zeros(nblk)
for k in order:
    d = rel[:, k] - last
    print(f"{k:2d} {names[k]:40s} mean {d.mean():9.0f}  med {np.median(d):9.0f}  cum {rel[:,k].mean():9.0f}")
    last = rel[:, k]
print(f"B1 head (stamp 20, from start): mean {rel[:,20].mean():.0f} med {np.median(rel[:,20]):.0f}")
print("block start spread (cycles):", (s[:,0].max()-s[:,0].min()), " end spread:", s[:,17].max()-s[:,17].min(), "total span", s[:,17].max()-s[:,0].min())
