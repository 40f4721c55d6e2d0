import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import geomae_oracle as O
from geomae_amd import synth, ops
import geomae_amd
from geomae_amd.configs import mae_sst_model
dev = torch.device('cuda:0')
RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
cfg = mae_sst_model(encoder_num_blocks=1, decoder_num_blocks=1); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev)
params = O.make_params(7, 1, 1); model.load_state_dict(params, strict=False); model.train()
bb = model.backbone
frames = [synth.lidar_frame(21), synth.lidar_frame(22, beams=16, n_az=300)]
_, coors = O.voxelize_batch(frames, (0.256, 0.256, 8), RANGE)
vc = torch.as_tensor(O.unique_rows(coors)[0], device=dev)
n = vc.shape[0]; print("n", n)
x0 = torch.randn(n, 128, generator=torch.Generator().manual_seed(0)).to(dev)
w = torch.randn(n, 128, generator=torch.Generator().manual_seed(1)).to(dev)
pg = []
for mode in ("bf16", "fp32"):
    bb.compute_dtype = mode
    for p in bb.parameters(): p.grad = None
    if bb.fused: bb._packed.refresh()
    layouts, pos = bb.get_voxel_info(vc, 2)
    x = x0.clone().requires_grad_(True)
    y = bb._run_stack(bb.encoder_blocks, "enc", x, pos, layouts)
    (y * w).sum().backward()
    pg.append({k: v.grad.float().cpu().numpy().copy() for k, v in bb.encoder_blocks.named_parameters()})
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
for k in pg[0]:
    r = rel(pg[0][k], pg[1][k])
    extra = ""
    if r > 4e-2 and pg[0][k].ndim == 1:
        ratio = pg[0][k] / np.where(np.abs(pg[1][k]) > 1e-6, pg[1][k], 1)
        extra = f" ratio med {np.median(ratio):.3f} first8 {np.round(ratio[:8],2)} last8 {np.round(ratio[-8:],2)}"
    print(f"{k:60s} {r:.4f}{extra}")
