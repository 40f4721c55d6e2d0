import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import geomae_oracle as O
from geomae_amd import ops, synth
from test_gpu_parity import _build, LEVELS, RANGE
dev=torch.device('cuda:0')
model,_=_build(dev,1,1,"bf16"); bb=model.backbone
frames=[synth.lidar_frame(31), synth.lidar_frame(32, beams=16, n_az=300)]
_, coors = O.voxelize_batch(frames, LEVELS["top"], RANGE)
vc = torch.as_tensor(O.unique_rows(coors)[0], device=dev)
n=vc.shape[0]
x0 = torch.randn(n, 128, generator=torch.Generator().manual_seed(0)).to(dev)
w = torch.randn(n, 128, generator=torch.Generator().manual_seed(1)).to(dev)
res={}
for mode in ("bf16","fp32"):
    bb.compute_dtype=mode
    for p in bb.parameters(): p.grad=None
    if bb.fused: bb._packed.refresh()
    layouts,pos=bb.get_voxel_info(vc,2)
    x=x0.clone().requires_grad_(True)
    y=bb._run_stack(bb.encoder_blocks,"enc",x,pos,layouts)
    print(mode,"fwd finite",bool(torch.isfinite(y).all()), "nan rows", int((~torch.isfinite(y)).any(1).sum()), "of", n)
    (y*w).sum().backward()
    print(mode,"dx finite",bool(torch.isfinite(x.grad).all()), "nan rows", int((~torch.isfinite(x.grad)).any(1).sum()))
    res[mode]=(y.detach(),x.grad)
a,b=res["bf16"],res["fp32"]
bad=(~torch.isfinite(a[0])).any(1).nonzero().flatten()
print("first bad rows", bad[:20].tolist())
err=(a[0]-b[0]).abs().max(1).values
print("rows with err>0.5:", int((err>0.5).sum()), (err>0.5).nonzero().flatten()[:20].tolist())
