"""Shape of ONE fused ffn-backward launch of the encoder (layer 0's: ffn workgroups + the contraction of layer 1 + the
reduction of layer 2): when each workgroup started and ended on the device-wide 100 MHz clock (s_memrealtime).
Build first:  GEOMAE_TIMING_DEFS="-DGEOMAE_STAMP_MAX_GRID=500" python tools/build_timing.py
(decoder size: GEOMAE_TIMING_DEFS="-DGEOMAE_STAMP_BLOCKS=1024 -DGEOMAE_STAMP_MIN_GRID=500", STAMP_BLOCKS=1024 launch_shape.py decoder)"""
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
SL, NB = 32, int(os.environ.get("STAMP_BLOCKS", "512"))


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
DEC = len(sys.argv) > 1 and sys.argv[1] == "decoder"          # all pillars through a decoder stack instead of the kept 30 %
keep = torch.rand(vc_all.shape[0], generator=torch.Generator().manual_seed(0)).to(dev) < (2.0 if DEC else 0.3)
vc = vc_all[keep].contiguous()
n = vc.shape[0]
x = torch.randn(n, 128, device=dev); dz = torch.randn(n, 128, device=dev)
bb._packed.refresh()
layouts, _ = bb.get_voxel_info(vc, len(pts))
blocks, key = (bb.decoder_centroid_blocks, "cen") if DEC else (bb.encoder_blocks, "enc")
nl = 2 * len(blocks)
w = bb._packed.weight_array(bb._stack_base[key], nl)
for p in bb.parameters():
    p.grad = None
g = bb._packed.grad_array(bb._stack_base[key], nl)
z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
for _ in range(3):
    ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
torch.cuda.synchronize(); read()
ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
torch.cuda.synchronize()
st = read()
ok = st[(st[:, 28] > 0) & (st[:, 29] > 0)]
t0 = ok[:, 28].min()
print(f"{n} tokens; {len(ok)} workgroups of the last fused launch; span {(ok[:, 29].max() - t0) / 100:.2f} us")
for kind, name in ((1, "ffn"), (2, "contraction"), (3, "reduction")):
    k = ok[ok[:, 30] == kind]
    if len(k):
        s, e = (k[:, 28] - t0) / 100.0, (k[:, 29] - t0) / 100.0
        print(f"  {name:12s} {len(k):4d} workgroups: start {s.min():5.2f} .. {s.max():5.2f} us (median {np.median(s):5.2f}), "
              f"end {e.min():5.2f} .. {e.max():5.2f} (median {np.median(e):5.2f}), duration median {np.median(e - s):5.2f} max {(e - s).max():5.2f}")
