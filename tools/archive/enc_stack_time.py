"""Encoder stack alone (forward, and forward + backward), event-timed, single-wave vs pair kernels.
usage: python tools/enc_stack_time.py [library.so]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
lib = _lib.load(path=sys.argv[1]) if len(sys.argv) > 1 else _lib.load()
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
x = torch.randn(n, 128, device=dev)
dz = torch.randn(n, 128, device=dev)
bb._packed.refresh()
layouts, _ = bb.get_voxel_info(vc, len(pts))
nl = 2 * len(bb.encoder_blocks)
w = bb._packed.weight_array(bb._stack_base["enc"], nl)
for p in bb.parameters():
    p.grad = None
g = bb._packed.grad_array(bb._stack_base["enc"], nl)
print(f"encoder stack alone: {n} tokens, {nl} layers", sys.argv[1:] or "product library")


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / reps


for mode in (0, 1, 0, 1):
    lib.geomae_sst_set_pair_kernels(mode)
    saved = [None]
    def fwd():
        saved[0] = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])[1]
    def both():
        fwd()
        ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved[0])
    f = timed(fwd)
    b = timed(both)
    print(f"pair={mode}: forward {f:7.1f} us   forward+backward {b:7.1f} us   (backward {b - f:7.1f})")
lib.geomae_sst_set_pair_kernels(-1)
