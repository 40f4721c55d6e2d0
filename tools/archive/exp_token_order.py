"""Experiment: does the token ORDER inside a stack matter?  Encoder / decoder stack forward+backward with the tokens in
pillar order (as produced by the pipeline) vs. permuted so that every unshifted window is a run of consecutive tokens."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import geomae_amd
from geomae_amd import ops, synth
from geomae_amd.configs import mae_sst_model
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
pts = [torch.as_tensor(synth.lidar_frame(10000 + b), device=dev) for b in range(4)]
voxels, coors, _, _ = model.voxelize_all(pts)
seg = ops.pillar_segment(coors, 4, model.grid_size)
vc = seg.voxel_coors[:seg.V].contiguous()
ik, im, _, _ = ops.random_mask(seg, 0.3, 1)
bb._packed.refresh()
nh, pt = bb.nhead[0], bb.pos_table


def run(name, c, stack, nl):
    n = c.shape[0]
    x = torch.randn(n, 128, device=dev)
    dz = torch.randn(n, 128, device=dev)
    L = ops.window_build_batch([(c, 0), (c, 1)], 4, bb._wcfg)
    w = bb._packed.weight_array(bb._stack_base[stack], nl)
    g = bb._packed.grad_array(bb._stack_base[stack], nl)
    def step():
        z, s = ops.sst_stack_forward(x, w, L, pt, nh)
        return ops.sst_stack_backward(dz, n, w, g, L, pt, nh, s)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    print(f"{name:34s} n={n:6d}  {1e3 * (time.perf_counter() - t0) / 30:.3f} ms fwd+bwd", flush=True)


for stack, c, nl in (("enc", vc[ik.long()].contiguous(), 12), ("cen", torch.cat([vc[ik.long()], vc[im.long()]]).contiguous(), 4)):
    L0 = ops.window_build(c, 4, bb._wcfg, 0)
    perm = L0.win_tokens[:c.shape[0]].long()
    for rep in range(2):
        run(f"{stack}: pillar order", c, stack, nl)
        run(f"{stack}: window-major order", c[perm].contiguous(), stack, nl)
