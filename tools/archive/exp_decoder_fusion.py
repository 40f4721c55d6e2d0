"""Experiment: the two 4-layer decoder stacks (n = all pillars of a B=4 batch).  (a) one stack alone, (b) the product
schedule: two stacks on two streams, (c) the timing of a horizontally fused launch sequence, emulated by ONE stack over
the token set duplicated as samples B..2B-1 (2n tokens, same kernels, twice the grid)."""
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
    vc = seg.voxel_coors[:V].contiguous()
    vc2 = torch.cat([vc, vc + torch.tensor([B, 0, 0, 0], dtype=torch.int32, device=dev)]).contiguous()
    P = bb._packed
    P.refresh()
    for p in bb.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    Wc, Wd = P.weight_array(12, 4), P.weight_array(16, 4)
    Gc, Gd = P.grad_array(12, 4), P.grad_array(16, 4)
    nh, pt = bb.nhead[0], bb.pos_table
    L1 = ops.window_build_batch([(vc, 0), (vc, 1)], B, bb._wcfg)
    L2 = ops.window_build_batch([(vc2, 0), (vc2, 1)], 2 * B, bb._wcfg)
    x1 = torch.randn(V, 128, device=dev)
    x2 = torch.cat([x1, x1])
side = torch.cuda.Stream()


def timeit(fn, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        keep = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts)[: reps // 2]
    return sum(ts) / len(ts)


def single_fwd():
    return ops.sst_stack_forward(x1, Wc, L1, pt, nh)


def pair_fwd():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    a = ops.sst_stack_forward(x1, Wd, L1, pt, nh, stream=side)
    b = ops.sst_stack_forward(x1, Wc, L1, pt, nh)
    cur.wait_stream(side)
    return a, b


def fused_fwd():
    return ops.sst_stack_forward(x2, Wc, L2, pt, nh)


print("forward  single %.4f  two-streams %.4f  fused-emulation %.4f ms" % (timeit(single_fwd), timeit(pair_fwd), timeit(fused_fwd)), flush=True)
z1, s1 = single_fwd()
(zd, sd), (zc, sc) = pair_fwd()
z2, s2 = fused_fwd()
dz1, dz2 = torch.randn_like(z1), torch.randn_like(z2)
torch.cuda.synchronize()


def single_bwd():
    return ops.sst_stack_backward(dz1, V, Wc, Gc, L1, pt, nh, s1)


def pair_bwd():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    a = ops.sst_stack_backward(dz1, V, Wd, Gd, L1, pt, nh, sd, stream=side)
    b = ops.sst_stack_backward(dz1, V, Wc, Gc, L1, pt, nh, sc)
    cur.wait_stream(side)
    return a, b


def fused_bwd():
    return ops.sst_stack_backward(dz2, 2 * V, Wc, Gc, L2, pt, nh, s2)


print("backward single %.4f  two-streams %.4f  fused-emulation %.4f ms" % (timeit(single_bwd), timeit(pair_bwd), timeit(fused_bwd)), flush=True)
