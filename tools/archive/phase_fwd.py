"""Per-phase cycles inside the FFN-forward kernel (single-wave and pair form) at encoder size, from clock64 stamps.
Build first:  GEOMAE_TIMING_DEFS="-DGEOMAE_STAMP_FWD -DGEOMAE_STAMP_MAX_GRID=300" python tools/build_timing.py
(the first FFN-forward launch after a clear keeps its stamps: encoder layer 0, which carries the next layer's QKV)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
lib = _lib.load(path=os.path.join(ROOT, "tools", "libgeomae_timing.so"))
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
from geomae_amd.train import Trainer
lib.geomae_debug_read_stamps.restype = ctypes.c_int
lib.geomae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
SL, NB = 32, 512
NAMES = ["loads + out-proj GEMM", "exchange/LN1/affine/pack", "GEMM1 (128->256)", "hp store + GELU (+exchange)",
         "GEMM2 (256->128)", "residual (+exchange) LN2 stores", "pos add, packs, x_b stores", "QK GEMM (128->256)",
         "qk store + V GEMM", "v store"]


def read():
    buf = np.zeros(NB * SL, dtype=np.uint64)
    lib.geomae_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(NB, SL).astype(np.int64)


dev = torch.device("cuda:0")
from geomae_amd import ops
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(4)]
_, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, len(pts), model.grid_size)
vc_all = seg.voxel_coors[:seg.V]
keep = torch.rand(vc_all.shape[0], generator=torch.Generator().manual_seed(0)).to(dev) < 0.3   # the kept 30 %
vc = vc_all[keep].contiguous()
n = vc.shape[0]
x = torch.randn(n, 128, device=dev)
bb._packed.refresh()
layouts, _ = bb.get_voxel_info(vc, len(pts))
nl = 2 * len(bb.encoder_blocks)
w = bb._packed.weight_array(bb._stack_base["enc"], nl)
print("encoder stack alone:", n, "tokens,", nl, "layers")
for mode in (0, 1):
    lib.geomae_sst_set_pair_kernels(mode)
    for _ in range(3):
        ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
    torch.cuda.synchronize(); read()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
    t1.record()
    torch.cuda.synchronize()
    st = read()
    nb = int((st[:, 0] > 0).sum())
    s = st[:nb]
    print(f"---- pair={mode}: stack forward {t0.elapsed_time(t1) * 1e3:.0f} us; {nb} workgroups stamped; kernel span "
          f"{s[:, 10].max() - s[:, 0].min()} cycles, start spread {s[:, 0].max() - s[:, 0].min()}")
    for k, name in enumerate(NAMES):
        d = s[:, k + 1] - s[:, k]
        print(f"  {name:36s} mean {d.mean():8.0f}  med {np.median(d):8.0f}")
    print(f"  {'total per workgroup':36s} mean {(s[:, 10] - s[:, 0]).mean():8.0f}")
    ok = s[(s[:, 10] > 0)]
    t0 = ok[:, 0].min()
    q = lambda v: " ".join(f"{int(x):7d}" for x in np.percentile(v, [0, 10, 25, 50, 75, 90, 100]))
    print("  workgroup start offsets (pct 0/10/25/50/75/90/100):", q(ok[:, 0] - t0))
    print("  workgroup end   offsets                           :", q(ok[:, 10] - t0))
