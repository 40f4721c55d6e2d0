"""Phase stamps of sst_ffn_fwd_kernel at decoder (or encoder) size: the stack forward alone in the three-launch form.
Build first:  GEOMAE_TIMING_DEFS=-DGEOMAE_STAMP_FWD python tools/build_timing.py      Usage: python tools/ffn_fwd_time.py [enc|dec]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "dec"
from geomae_amd import _lib
lib = _lib.load(path=os.path.join(ROOT, "tools", "libgeomae_timing.so"))
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
NAMES = ["rows + Wo GEMM + residual", "LN1 (+ xhat store)", "W1 GEMM", "hp store + GELU", "W2 GEMM", "LN2 + stores (xhat2, rstd, z)",
         "next layer: pos rows, x / xp stores", "next q k GEMM", "qk store (+ Wv issue)", "next v GEMM + store"]
def read():
    buf = np.zeros(512 * 32, dtype=np.uint64)
    lib.geomae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.geomae_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(512, 32).astype(np.int64)
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
pts = [torch.as_tensor(synth.lidar_frame(10000 + b, sweeps=int(os.environ.get("SWEEPS", "1"))), device=dev) for b in range(4)]
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
layouts, _ = bb.get_voxel_info(vc, len(pts))
nl = 2 * len(blocks)
w = bb._packed.weight_array(bb._stack_base[name], nl)
lib.geomae_sst_set_fused_layers(0)
for _ in range(3): ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
torch.cuda.synchronize(); read()
ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])          # (the first ffn launch after the clear keeps its stamps)
st = read(); s = st[st[:, 10] > 0]
print(f"{which}: {n} tokens, {nl} layers; sst_ffn_fwd_kernel, first launch of the stack: {len(s)} workgroups stamped (wave 0), "
      f"total mean {(s[:, 10] - s[:, 0]).mean():.0f} max {(s[:, 10] - s[:, 0]).max()} cycles")
for k, nm in enumerate(NAMES):
    d = s[:, k + 1] - s[:, k]
    print(f"    {nm:40s} mean {d.mean():8.0f}  max {d.max():8.0f}")
