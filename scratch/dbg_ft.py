import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import geomae_oracle as O
import geomae_amd
from geomae_amd import ops
dev = torch.device('cuda:0')
RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
FT_DROP = {0: dict(max_tokens=30, drop_range=(0, 30)), 1: dict(max_tokens=60, drop_range=(30, 60)), 2: dict(max_tokens=144, drop_range=(60, 100000))}
g = np.load('/root/repo/tests/golden/g_finetune.npz')
def run(tag, contiguous=False):
    mid = geomae_amd.SSTInputLayer(drop_info=(FT_DROP, FT_DROP), shifts_list=[(0, 0), (6, 6)], window_shape=(12, 12), point_cloud_range=RANGE, voxel_size=(0.256, 0.256, 8), shuffle_voxels=False, debug=False).to(dev)
    bb = geomae_amd.SSTSecondPretrainedv1(d_model=[128, 128], nhead=[8, 8], num_blocks=1, dim_feedforward=[256, 256], output_shape=[400, 400], conv_in_channels=128, conv_out_channels=[32, 48], layer_nums=[1, 2], layer_strides=[2, 2], debug=False, drop_info=(FT_DROP, FT_DROP), window_shape=(12, 12), compute_dtype="fp32").to(dev)
    bb.load_state_dict(O.seeded_state(5, {k: v.shape for k, v in bb.state_dict().items()}))
    mid.train(); bb.train()
    vc = torch.as_tensor(g["coors"].astype(np.int32), device=dev); n = int(g["n"])
    x = torch.randn(n, 128, generator=torch.Generator().manual_seed(3)).to(dev).requires_grad_(True)
    if contiguous:
        orig = ops.recover_bev
        ops.recover_bev = lambda *a: orig(*a).contiguous()
    outs = bb(mid(x, vc, 2))
    if contiguous: ops.recover_bev = orig
    w = [torch.randn(tuple(int(v) for v in g[f"out{i}_shape"]), generator=torch.Generator().manual_seed(11 + i)).to(dev) for i in range(len(outs))]
    loss = sum((o * wi).sum() for o, wi in zip(outs, w)) * 1e-2
    loss.backward()
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
    print(tag, "loss", float(loss), "ref", float(g["loss"]), "dx rel", rel(x.grad.cpu().numpy(), g["dx"]),
          "out abs rel", [abs(float(o.double().abs().sum()) - float(g[f"out{i}_abs"])) / float(g[f"out{i}_abs"]) for i, o in enumerate(outs)])
    gn = {k: float(p.grad.double().norm()) for k, p in bb.named_parameters()}
    bad = [(str(k), gn[str(k)], float(r)) for k, r in zip(g["grad_names"], g["grad_norms"]) if abs(gn[str(k)] - r) > 0.01 * max(r, 1e-6)]
    print("   bad grads", bad[:6], len(bad))
print("allow_tf32", torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32)
run("default")
torch.backends.cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
run("no tf32")
run("contiguous canvas", True)
torch.backends.cudnn.deterministic = True
run("deterministic")
