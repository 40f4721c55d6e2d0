import os, sys
import torch
sys.path.insert(0, '/root/repo')
from geomae_amd import _lib
lib = _lib.load()
import geomae_amd
from geomae_amd import ops, synth
from geomae_amd.configs import mae_sst_model
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
for p in bb.parameters(): p.grad = None
g = bb._packed.grad_array(bb._stack_base["enc"], nl)
z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ta = tb = 0.0
R = 20
for i in range(R + 3):
    ev[0].record()
    dx, scratch = ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved, defer_last=True)
    ev[1].record()
    ops.flush_weight_grad()
    ev[2].record()
    torch.cuda.synchronize()
    if i >= 3:
        ta += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"n={n}: backward without the last dW {ta / R * 1e3:.1f} us, the deferred dW(0) + reduction alone {tb / R * 1e3:.1f} us")
