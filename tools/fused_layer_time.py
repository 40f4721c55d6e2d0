"""The encoder / decoder stack forward alone, one-launch layers (csrc/sst_fused.hip) against the three-launch form, and the
per-phase cycles inside the one-launch kernel from clock64 stamps (wave 0 and wave 4 of every workgroup).
Build first:  python tools/build_timing.py    Usage: python tools/fused_layer_time.py [enc|dec] [cap]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "enc"
if len(sys.argv) > 2:
    os.environ["GEOMAE_BUNDLE_CAP"] = sys.argv[2]
from geomae_amd import _lib
timing = os.environ.get("GEOMAE_TIMING_LIB") or os.path.join(ROOT, "tools", "libgeomae_timing.so")
lib = _lib.load(path=timing if os.path.exists(timing) and not os.environ.get("NO_STAMPS") else None)
import geomae_amd
from geomae_amd import synth, ops
from geomae_amd.configs import mae_sst_model
has_stamps = hasattr(lib, "geomae_debug_read_fused_stamps")
SL, NBLK = 32, 512
NAMES = ["A: plan, rows, pos -> LDS", "barrier 1", "B: q k v projection", "C: attention", "barrier 2", "D: out-proj + partial LN",
         "barrier 3", "LN1 -> y", "barrier 4", "E: FFN1 + gelu", "barrier 5", "F: FFN2 + partial LN", "barrier 6", "LN2, z"]


def read():
    buf = np.zeros(NBLK * SL, dtype=np.uint64)
    lib.geomae_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(NBLK, SL).astype(np.int64)


dev = torch.device("cuda:0")
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
SWEEPS = int(os.environ.get("SWEEPS", "1"))
pts = [torch.as_tensor(synth.lidar_frame(10000 + b, sweeps=SWEEPS), device=dev) for b in range(4)]
_, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, len(pts), model.grid_size)
ids_keep, ids_mask, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1, bb._wcfg)      # window-major, as in a step
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
if os.environ.get("SAME_WEIGHTS"):          # every layer reads layer 0's weights: are they still in L2 at the next launch?
    from geomae_amd._lib import GeomaeSstLayerWeights
    w = (GeomaeSstLayerWeights * nl)(*[bb._packed.structs[bb._stack_base[name]]] * nl)
nb = [int(L.num_fbundles.item()) for L in layouts]
sz = [(L.fbun_tok[1:b + 1] - L.fbun_tok[:b]).cpu().numpy() for L, b in zip(layouts, nb)]
print(f"{which} stack alone: {n} tokens, {nl} layers; bundles {nb}, tokens per bundle mean {sz[0].mean():.1f} max {sz[0].max()} / {sz[1].max()}, "
      f"tiles {sum(((s + 15) // 16).sum() for s in sz) / 2:.0f} for {n / 16:.0f}")
for mode in (0, 1):
    lib.geomae_sst_set_fused_layers(mode)
    for _ in range(3):
        ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
    torch.cuda.synchronize()
    if has_stamps:
        read()
    ts = []
    for _ in range(5):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
        t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3)
    print(f"---- one-launch={mode}: stack forward {np.median(ts):.0f} us ({np.median(ts) / nl:.1f} us per layer)")
    if mode == 1 and has_stamps:
        st = read()                      # (the LAST layer's launch overwrote the others)
        if os.environ.get("DUMP_WG"):
            s0 = st[:, 0:16]; ok = s0[:, 0] > 0
            tot = (s0[:, 14] - s0[:, 0])[ok]; a = (s0[:, 1] - s0[:, 0])[ok]
            T = sz[(nl - 1) & 1]; T = np.concatenate([T, -np.ones(max(0, ok.sum() - len(T)), dtype=T.dtype)])[:ok.sum()]
            order = np.argsort(tot)
            print("  slowest workgroups (block, tokens, total, phase A, start offset):")
            t0 = s0[ok][:, 0].min()
            for i in list(order[-12:]) + list(order[:6]):
                print(f"    b={i:4d} T={T[i] if i < len(T) else -1:3d} total={tot[i]:6d} A={a[i]:6d} start={s0[ok][i, 0] - t0:6d} xcd={i % 8}")
            for ntv in range(1, 6):
                m = ((T + 15) // 16) == ntv
                if m.any(): print(f"    nt={ntv}: {m.sum():3d} workgroups, total mean {tot[m].mean():.0f} max {tot[m].max()}")
        for half in (0, 1):
            s = st[:, 16 * half:16 * half + 16]
            s = s[s[:, 0] > 0]
            print(f"  wave {4 * half}: {len(s)} workgroups, total per workgroup mean {(s[:, 14] - s[:, 0]).mean():.0f} max {(s[:, 14] - s[:, 0]).max()} cycles")
            for k, nm in enumerate(NAMES):
                d = s[:, k + 1] - s[:, k]
                print(f"    {nm:28s} mean {d.mean():8.0f}  med {np.median(d):8.0f}  max {d.max():8.0f}")
