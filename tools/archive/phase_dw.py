"""Cycles inside the weight-gradient contraction workgroups that ride in the encoder's ffn-backward launches (clock64 stamps).
Build first:  GEOMAE_TIMING_DEFS="-DGEOMAE_STAMP_MAX_GRID=300" python tools/build_timing.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
lib = _lib.load(path=os.path.join(ROOT, "tools", "libgeomae_timing.so"))
import geomae_amd
from geomae_amd import ops, synth
from geomae_amd.configs import mae_sst_model
lib.geomae_debug_read_stamps.restype = ctypes.c_int
lib.geomae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
SL, NB = 32, 512


def read():
    buf = np.zeros(NB * SL, dtype=np.uint64)
    lib.geomae_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(NB, SL).astype(np.int64)


dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(4)]
_, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, len(pts), model.grid_size)
vc_all = seg.voxel_coors[:seg.V]
keep = torch.rand(vc_all.shape[0], generator=torch.Generator().manual_seed(0)).to(dev) < 0.3
vc = vc_all[keep].contiguous()
n = vc.shape[0]
x = torch.randn(n, 128, device=dev); dz = torch.randn(n, 128, device=dev)
bb._packed.refresh()
layouts, _ = bb.get_voxel_info(vc, len(pts))
nl = 2 * len(bb.encoder_blocks)
w = bb._packed.weight_array(bb._stack_base["enc"], nl)
for p in bb.parameters():
    p.grad = None
g = bb._packed.grad_array(bb._stack_base["enc"], nl)
z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
for _ in range(3):
    ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
torch.cuda.synchronize(); read()
ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
torch.cuda.synchronize()
st = read()          # the LAST launch that stamped: layer 0's ffn-backward (+ dW of layer 1 + reduction of layer 2)
ffn = st[(st[:, 0] > 0) & (st[:, 17] > 0)]
dw = st[(st[:, 24] > 0) & (st[:, 26] > 0)]
print(f"{n} tokens; ffn workgroups stamped {len(ffn)}: total {np.median(ffn[:, 17] - ffn[:, 0]):.0f} cycles (median)")
print(f"dW workgroups stamped {len(dw)}: loop {np.median(dw[:, 25] - dw[:, 24]):.0f}, partial stores / atomics issue "
      f"{np.median(dw[:, 26] - dw[:, 25]):.0f}, total {np.median(dw[:, 26] - dw[:, 24]):.0f} cycles (median); "
      f"max total {np.max(dw[:, 26] - dw[:, 24])}")
