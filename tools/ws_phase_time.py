"""Phase cycles inside sst_layer_fwd_ws_kernel (csrc/sst_ws.hip): clock64 stamps of wave 0 over every workgroup's FIRST bundle,
grouped by the bundle's tile count.  Build first: python tools/build_timing.py.   Usage: python tools/ws_phase_time.py [enc|dec] [cap]
(SWEEPS=10: config 3's sizes)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "dec"
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 144
from geomae_amd import _lib
lib = _lib.load(path=os.path.join(ROOT, "tools", "libgeomae_timing.so"))
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
SL, NBLK = 32, 512
NAMES = ["A: rows, pos -> LDS", "barrier 1", "B: k v projection", "C: q + attention", "barrier 2", "D: x re-read, out-proj", "barrier 3",
         "LN1 -> y", "barrier 4", "E: FFN1 + gelu", "barrier 5", "F: FFN2", "barrier 6", "LN2, z", "(further groups)"]
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
SWEEPS = int(os.environ.get("SWEEPS", "1"))
pts = [torch.as_tensor(synth.lidar_frame(10000 + b, sweeps=SWEEPS), device=dev) for b in range(4)]
_, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, len(pts), model.grid_size)
ids_keep, ids_mask, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1, bb._wcfg)
vc_all = seg.voxel_coors[:seg.V]
if which == "enc":
    vc = vc_all[ids_keep.long()].contiguous(); name, blocks = "enc", bb.encoder_blocks
else:
    vc = torch.cat([vc_all[ids_keep.long()], vc_all[ids_mask.long()]]).contiguous(); name, blocks = "cen", bb.decoder_centroid_blocks
n = vc.shape[0]
x = torch.randn(n, 128, device=dev)
bb._packed.refresh()
nl = 2 * len(blocks)
w = bb._packed.weight_array(bb._stack_base[name], nl)
_lib.set_tuning(ws_layers=2, fused_layers=1, ws_bundle_cap=cap, bundle_cap=cap)
layouts, _ = bb.get_voxel_info(vc, len(pts))
for _ in range(3):
    ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
buf = np.zeros(NBLK * SL, dtype=np.uint64)
lib.geomae_debug_read_ws_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
lib.geomae_debug_read_ws_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
st = buf.reshape(NBLK, SL).astype(np.int64)
st = st[st[:, 0] > 0]
T = st[:, 16]
nt = (T + 15) // 16
print(f"{which}: {n} tokens, cap {cap}; {len(st)} workgroups stamped (first bundle of each; the LAST layer's launch)")
for v in sorted(set(nt)):
    s = st[nt == v]
    tot = s[:, 15] - s[:, 0]
    print(f"-- bundles of {v} tiles: {len(s)} workgroups, first bundle total mean {tot.mean():.0f} max {tot.max()} cycles")
    for k, nm in enumerate(NAMES):
        d = s[:, k + 1] - s[:, k]
        print(f"    {nm:26s} mean {d.mean():8.0f}  med {np.median(d):8.0f}  max {d.max():8.0f}")
