"""Experiment: the 12-layer encoder stack on all kept tokens of a B=4 batch in one chain vs. split by samples into K
independent chains on K streams (windows never cross samples).  Forward and backward, HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import geomae_amd
from geomae_amd import ops, synth
from geomae_amd.configs import mae_sst_model

dev = torch.device("cuda:0")
torch.manual_seed(1)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
B = 4
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(B)]
with torch.no_grad():
    voxels, coors, _, _ = model.voxelize_all(pts)
    seg = ops.pillar_segment(coors, B, model.grid_size)
    V = seg.V
    starts = seg.sync_counts()
    ik, im, token_row, counts = ops.random_mask(seg, 0.3, 1)
    vc = seg.voxel_coors[:V]
    keep_coors = vc[ik.long()].contiguous()
    bidx = keep_coors[:, 0]
    vf = torch.randn(V, 128, device=dev)
    P = bb._packed
    P.refresh()
    for p in bb.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    W = P.weight_array(0, 12)
    G = P.grad_array(0, 12)
    nh, pt = bb.nhead[0], bb.pos_table


def run(K, reps=30):
    # split points: sample boundaries
    per = [int((bidx < b).sum()) for b in range(B + 1)]
    cuts = [per[(B * k) // K] for k in range(K + 1)]
    # side streams on hardware queues of their own (ops.side_streams: measured), not just any new streams
    st = ops.side_streams(dev)
    side = [st["geo"], st["dec_b"]] + [torch.cuda.Stream() for _ in range(max(0, K - 3))]
    streams = [torch.cuda.current_stream()] + side[:K - 1]
    parts = []
    for k in range(K):
        a, b = cuts[k], cuts[k + 1]
        c = keep_coors[a:b].contiguous()
        lay = [ops.window_build(c, B, bb._wcfg, s) for s in (0, 1)]
        # attention plan
        L = ops.window_build_batch([(c, 0), (c, 1)], B, bb._wcfg)
        parts.append((a, b, L, ik[a:b].contiguous()))
    torch.cuda.synchronize()
    res = {}
    for phase in ("fwd", "bwd"):
        times = []
        for r in range(reps):
            saved = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = streams[0]
            # forward (always; timed only in fwd phase)
            if phase == "fwd":
                e0.record()
            for s in streams[1:]:
                s.wait_stream(cur)
            outs = []
            for (a, b, L, rows), s in zip(parts, streams):
                z, sv = ops.sst_stack_forward(vf, W, L, pt, nh, stream=None if s is cur else s, rows=rows)
                outs.append((z, sv))
            for s in streams[1:]:
                cur.wait_stream(s)
            if phase == "fwd":
                e1.record()
            else:
                dzs = [torch.randn_like(z) for z, _ in outs]
                d_vf = torch.zeros(V, 128, device=dev)
                torch.cuda.synchronize()
                e0.record()
                for s in streams[1:]:
                    s.wait_stream(cur)
                keep = []
                for (a, b, L, rows), s, (z, sv), dz in zip(parts, streams, outs, dzs):
                    r_ = ops.sst_stack_backward(dz, b - a, W, G, L, pt, nh, sv, stream=None if s is cur else s,
                                                scatter=(rows, d_vf))
                    keep.append(r_)
                for s in streams[1:]:
                    cur.wait_stream(s)
                e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        times = sorted(times)[: max(1, len(times) // 2)]
        res[phase] = sum(times) / len(times)
    return res


for K in (1, 2, 3, 4):
    print(K, "chains:", {k: round(v, 4) for k, v in run(K).items()}, "ms", flush=True)
