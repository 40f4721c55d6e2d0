"""Tiles per bundle of the one-launch layer kernel's packing (fbun_tok) at the encoder, per frame batch and window shift.
Usage: python tools/bundle_hist.py [batches]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    pts = [torch.as_tensor(synth.lidar_frame(10000 + 4 * k + b), device=dev) for b in range(4)]
    _, coors, _, _ = model.voxelize_all(pts)
    seg = ops.pillar_segment(coors, len(pts), model.grid_size)
    ids_keep, _, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1 + k, bb._wcfg)
    vc = seg.voxel_coors[:seg.V][ids_keep.long()].contiguous()
    layouts, _ = bb.get_voxel_info(vc, len(pts))
    for s, L in enumerate(layouts):
        nb = int(L.num_fbundles.item())
        bt = L.fbun_tok[:nb + 1].cpu().numpy()
        tiles = (np.diff(bt) + 15) // 16
        wt = np.diff(L.win_start[:int(L.num_windows.item()) + 1].cpu().numpy()) if hasattr(L, "win_start") else np.zeros(1)
        print(f"batch {k} shift {s}: {vc.shape[0]} tokens, {nb} bundles, tiles per bundle {dict(zip(*np.unique(tiles, return_counts=True)))}, "
              f"windows with > 48 tokens: {(wt > 48).sum()}, > 64: {(wt > 64).sum()}, max {wt.max()}")
