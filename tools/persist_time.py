"""The encoder stack forward alone (config 2's kept pillars, 12 layers): three launches per layer / one launch per layer /
ONE persistent launch for the stack (csrc/sst_fused.hip sst_stack_fwd_kernel).  Usage: python tools/persist_time.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
timing = os.path.join(ROOT, "tools", "libgeomae_timing.so")
lib = _lib.load(path=timing if os.path.exists(timing) and os.environ.get("STAMPS") else None)
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(4)]
_, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, len(pts), model.grid_size)
ids_keep, ids_mask, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1, bb._wcfg)
vc = seg.voxel_coors[:seg.V][ids_keep.long()].contiguous()
n = vc.shape[0]
x = torch.randn(n, 128, device=dev)
bb._packed.refresh()
layouts, _ = bb.get_voxel_info(vc, len(pts))
nl = 2 * len(bb.encoder_blocks)
w = bb._packed.weight_array(bb._stack_base["enc"], nl)
outs = {}
for mode, label in ((0, "three launches per layer"), (3, "one launch per layer"), (1, "one persistent launch")):
    lib.geomae_sst_set_fused_layers(mode)
    for _ in range(3):
        z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
        t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3)
    outs[mode] = z.clone()
    print(f"{label:28s}: {n} tokens x {nl} layers: {np.median(ts):.0f} us ({np.median(ts) / nl:.1f} us per layer; min {min(ts):.0f})", flush=True)
print("max |z| difference persistent vs per-layer:", float((outs[1] - outs[3]).abs().max()), " vs three-launch:", float((outs[1] - outs[0]).abs().max()))

if hasattr(lib, "geomae_debug_read_persist_stamps"):
    import ctypes
    lib.geomae_sst_set_fused_layers(1)
    ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
    buf = np.zeros(256 * 40, dtype=np.uint64)
    lib.geomae_debug_read_persist_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    st = buf.reshape(256, 40).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    print(f"{len(st)} workgroups; times in us from the first workgroup's start (s_memrealtime, 10 ns)")
    for l in range(nl):
        a, b, c = st[:, 3 * l], st[:, 3 * l + 1], st[:, 3 * l + 2]
        print(f"layer {l:2d}: start {0.01 * (a.min() - t0):7.2f} .. {0.01 * (a.max() - t0):7.2f} | bundles done: first {0.01 * (b.min() - t0):7.2f} "
              f"median {0.01 * (np.median(b) - t0):7.2f} last {0.01 * (b.max() - t0):7.2f} | body mean {0.01 * (b - a).mean():6.2f} max {0.01 * (b - a).max():6.2f} | "
              f"behind barrier: first {0.01 * (c.min() - t0):7.2f} last {0.01 * (c.max() - t0):7.2f}  (barrier after last arrival {0.01 * (c.max() - b.max()):5.2f})")
