"""The fused VFE alone: per-kernel times (rocprofv3-free: HIP events around forward / backward) and, with the timing build
(python tools/build_timing.py), the cycle split of the layer-1 forward sweep's point loop.
Usage: python tools/vfe_time.py [sweeps ...]     (default: 1 10 = BASELINE configs 2 and 3, 4 frames)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomae_amd import _lib
TIMING = os.path.exists(os.path.join(ROOT, "tools", "libgeomae_timing.so")) and not os.environ.get("GEOMAE_PRODUCT_LIB")
lib = _lib.load(path=os.path.join(ROOT, "tools", "libgeomae_timing.so")) if TIMING else _lib.load()
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
SL, NBLK = 32, 512
NAMES = ["staging", "features + layer 0 + m0 gather", "split + layer-1 GEMM issue", "BN/ReLU + tile -> LDS", "segmented max (scalar-steered walk)", "(loop back)"]
NAMES_B = ["staging", "features + layer 0 + m0 gather + g store", "GEMM 1 + routing gathers + dy1 stores", "GEMM 2 (W1^T) + dh0 stores + tile", "segmented sum"]
NAMES_R = ["staging", "features + layer 0", "routing (m0 / dm0 / dh0 gathers)", "tile + features -> LDS", "contraction MFMAs", "(loop back)", "after the loop"]
def read():
    buf = np.zeros(NBLK * SL, dtype=np.uint64)
    lib.geomae_debug_read_vfe_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.geomae_debug_read_vfe_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(NBLK, SL).astype(np.int64)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
ve = model.voxel_encoder
for sweeps in [int(a) for a in sys.argv[1:]] or [1, 10]:
    pts = [torch.as_tensor(synth.lidar_frame(10000 + b, sweeps=sweeps), device=dev) for b in range(4)]
    with torch.no_grad():
        voxels, top, med, low = model.voxelize_all(pts)
        seg = ops.pillar_segment(top, len(pts), model.grid_size)
        prepared = ve.prepare_points(voxels, seg)
        vf, state = ve.forward_explicit(voxels, seg, prepared=prepared)
        dvf = torch.randn_like(vf)
        for p in ve.parameters(): p.grad = None
        t_prep = timed(lambda: ve.prepare_points(voxels, seg))
        t_fwd = timed(lambda: ve.forward_explicit(voxels, seg, prepared=prepared))
        t_bwd = timed(lambda: ve.backward_explicit(state, dvf))
    print("pillars marked as possible ties:", int(state[0].pillar_ties.sum().item()))
    print(f"sweeps {sweeps}: {voxels.shape[0]} points, {seg.V} pillars: prepare {t_prep:.0f} us, forward sweeps {t_fwd:.0f} us, backward {t_bwd:.0f} us")
    if TIMING:
        read()
        with torch.no_grad(): ve.forward_explicit(voxels, seg, prepared=prepared)
        st = read(); s = st[st[:, 7] > 0]
        print(f"  vfe_layer1_kernel: {len(s)} workgroups stamped (wave 0), whole kernel mean {s[:, 7].mean():.0f} max {s[:, 7].max()} cycles")
        for k, nm in enumerate(NAMES): print(f"    {nm:36s} mean {s[:, k].mean():8.0f}  max {s[:, k].max():8.0f}")
        for p in ve.parameters(): p.grad = None
        with torch.no_grad(): ve.backward_explicit(state, dvf)
        st = read(); s = st[st[:, 7] > 0]
        print(f"  vfe_bwd_layer1_kernel: {len(s)} workgroups stamped (wave 0), whole kernel mean {s[:, 7].mean():.0f} max {s[:, 7].max()} cycles")
        for k, nm in enumerate(NAMES_B): print(f"    {nm:36s} mean {s[:, k].mean():8.0f}  max {s[:, k].max():8.0f}")
        try:
            lib.geomae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
            b2 = np.zeros(NBLK * SL, dtype=np.uint64)
            lib.geomae_debug_read_stamps(b2.ctypes.data_as(ctypes.c_void_p), 1)
            d = b2.reshape(NBLK, SL).astype(np.int64); d = d[d[:, 26] > 0]
            print(f"  dw_kernel (dW1 = dy1^T g): {len(d)} workgroups stamped, loop mean {(d[:, 25] - d[:, 24]).mean():.0f} max {(d[:, 25] - d[:, 24]).max()} cycles, "
                  f"epilogue mean {(d[:, 26] - d[:, 25]).mean():.0f}")
        except Exception as ex:
            print("  (no dw stamps:", ex, ")")
        s = st[st[:, 23] > 0]
        print(f"  vfe_bwd_stats1_kernel: {len(s)} workgroups stamped (wave 0), whole kernel mean {s[:, 23].mean():.0f} max {s[:, 23].max()} cycles")
        for k, nm in enumerate(["staging", "gathers + features + layer 0 (both passes)", "GEMM half + routing (both passes)", "flush of the channel sums"]):
            kk = [0, 1, 2, 3][k]
            print(f"    {nm:44s} mean {s[:, 16 + kk].mean():8.0f}  max {s[:, 16 + kk].max():8.0f}")
        s = st[st[:, 15] > 0]
        print(f"  vfe_bwd_route0_kernel<true>: {len(s)} workgroups stamped (wave 0), whole kernel mean {s[:, 15].mean():.0f} max {s[:, 15].max()} cycles")
        for k, nm in enumerate(NAMES_R): print(f"    {nm:36s} mean {s[:, 8 + k].mean():8.0f}  max {s[:, 8 + k].max():8.0f}")
