"""The window-attention kernels alone (row-major C-ABI calls) at encoder and decoder size of BASELINE config 2:
microseconds per launch, forward and backward.   python tools/attn_time.py [SWEEPS]   (LIB=tools/libgeomae_timing.so)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
_lib.load(path=os.environ.get("LIB") or None)
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pts = [torch.as_tensor(synth.lidar_frame(10000 + b, sweeps=sweeps), device=dev) for b in range(4)]
_, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, len(pts), model.grid_size)
ids_keep, ids_mask, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1, bb._wcfg)
vc_all = seg.voxel_coors[:seg.V]
for tag, vc in (("encoder", vc_all[ids_keep.long()].contiguous()), ("decoder", torch.cat([vc_all[ids_keep.long()], vc_all[ids_mask.long()]]).contiguous())):
    n = vc.shape[0]
    layouts, _ = bb.get_voxel_info(vc, len(pts))
    for s, L in enumerate(layouts):
        qkv = (torch.randn(n, 384, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
        dout = (torch.randn(n, 128, device=dev) * 0.5).to(torch.bfloat16)
        res = {}
        for what in ("fwd", "bwd"):
            ts = []
            for rep in range(8):
                qkv.grad = None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if what == "fwd":
                    e0.record(); out = ops.window_attention(qkv, L, 8); e1.record()
                else:
                    out = ops.window_attention(qkv, L, 8)
                    e0.record(); out.backward(dout); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res[what] = np.median(ts[2:])
        print(f"{tag} shift {s}: {n} tokens, {int(L.num_bundles.item())} bundles: fwd {res['fwd']:.1f} us  bwd {res['bwd']:.1f} us (incl. torch op overhead)")
