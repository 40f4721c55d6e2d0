"""The encoder stack forward alone over the one-launch forward's WORK ITEMS (GeomaeTuning.fwd_item_cap = 0 / 32 / 48 / 64; window.hip
item_pack) for four batches of four frames: item counts, how many are split parts, us per layer.  Usage: python tools/fwd_items_time.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geomae_amd import _lib
lib = _lib.load()
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
for base in (10000, 10004, 10008, 10012):
    pts = [torch.as_tensor(synth.lidar_frame(base + b), device=dev) for b in range(4)]
    _, coors, _, _ = model.voxelize_all(pts)
    seg = ops.pillar_segment(coors, len(pts), model.grid_size)
    ids_keep, ids_mask, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1, bb._wcfg)
    vc = seg.voxel_coors[:seg.V][ids_keep.long()].contiguous()
    n = vc.shape[0]
    x = torch.randn(n, 128, device=dev)
    bb._packed.refresh()
    w = bb._packed.weight_array(bb._stack_base["enc"], 12)
    for cap in (0, 32, 48, 64):
        _lib.set_tuning(fwd_item_cap=cap)
        layouts, _ = bb.get_voxel_info(vc, len(pts))
        desc = []
        for L in layouts:
            ni = int(L.num_fitems.item()); it = L.fitems[:ni].cpu().numpy(); nts = (it[:, 1] + 15) // 16
            desc.append(f"items {ni} split parts {(it[:,3] < nts).sum()} unsplit nt hist {np.bincount(nts[it[:,3]==nts], minlength=6)[1:6]}")
        for _ in range(3): ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(); ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0]); t1.record(); torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1) * 1e3)
        print(f"batch {base} n {n} fwd_item_cap {cap}: {np.median(ts):.0f} us ({np.median(ts)/12:.1f} per layer)  " + " | ".join(desc))
