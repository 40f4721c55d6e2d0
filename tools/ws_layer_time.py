"""A layer stack's forward (and forward + backward) alone: the three-launch form against the weight-stationary one-launch
layer (csrc/sst_ws.hip) at several bundle caps / workgroup counts.
Usage: python tools/ws_layer_time.py [enc|dec] [caps, e.g. 48,64,96,144]      (SWEEPS=10: config 3's sizes; BWD=1: + backward)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "dec"
caps = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "48,64,96,144").split(",")]
from geomae_amd import _lib
lib = _lib.load()
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model

dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
SWEEPS = int(os.environ.get("SWEEPS", "1"))
BWD = bool(os.environ.get("BWD"))
WGS = [int(v) for v in os.environ.get("WGS", "0").split(",")]
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
dz = torch.randn(n, 128, device=dev)
bb._packed.refresh()
nl = 2 * len(blocks)
w = bb._packed.weight_array(bb._stack_base[name], nl)
g = bb._packed.grad_array(bb._stack_base[name], nl)


def run(layouts, reps=7):
    def once():
        z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
        if BWD:
            ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
        return z
    for _ in range(3):
        z = once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); once(); t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3)
    return float(np.median(ts)), z


print(f"{which} stack alone: {n} tokens, {nl} layers{' (forward + backward)' if BWD else ''}")
_lib.set_tuning(ws_layers=0, fused_layers=0)
layouts, _ = bb.get_voxel_info(vc, len(pts))
t, z0 = run(layouts)
print(f"three-launch form            : {t:7.0f} us  ({t / nl:6.1f} us per layer)")
for cap in caps:
    for wg in WGS:
        _lib.set_tuning(ws_layers=2, fused_layers=1, ws_bundle_cap=cap, bundle_cap=cap, ws_max_workgroups=wg)
        layouts, _ = bb.get_voxel_info(vc, len(pts))
        nb = [int(L.num_fbundles.item()) for L in layouts]
        sz = (layouts[0].fbun_tok[1:nb[0] + 1] - layouts[0].fbun_tok[:nb[0]]).cpu().numpy()
        t, z1 = run(layouts)
        err = float((z1 - z0).norm() / z0.norm())
        print(f"weight-stationary cap {cap:3d} wg {wg:3d}: {t:7.0f} us  ({t / nl:6.1f} us per layer)  bundles {nb}, tiles per bundle "
              f"mean {((sz + 15) // 16).mean():.2f} max {((sz + 15) // 16).max()}, |z - z3| / |z3| = {err:.2e}")
