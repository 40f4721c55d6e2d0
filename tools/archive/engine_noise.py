import copy, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_engine import _build, _batches, _rel
from geomae_amd.train import Trainer
from geomae_amd.engine import PretrainEngine
m0 = _build()
pts = _batches(1)[0]
res = []
for kind in ("py", "py", "c", "c"):
    m = copy.deepcopy(m0)
    tr = Trainer(m)
    tr.flat.zero_grad()
    if kind == "py":
        l = m.train_step_explicit(pts)
        l = torch.stack([l[k] for k in m.LOSS_KEYS])
    else:
        eng = PretrainEngine(m, tr.flat, tr.opt, 10.0)
        l, _ = eng.step(pts, None, 1e-5, run_optimizer=False)
    torch.cuda.synchronize()
    res.append((kind, l.clone(), tr.flat.grad.clone(), tr))
for i in range(len(res)):
    for j in range(i + 1, len(res)):
        a, b = res[i], res[j]
        worst = max((_rel(a[2][o:o + p.numel()], b[2][o:o + p.numel()]), n) for n, o, p in zip(a[3].flat.names, a[3].flat.offsets, a[3].flat.params))
        print(a[0], b[0], "loss rel", float(((a[1] - b[1]).abs() / b[1].abs()).max()), "grad worst", worst, "whole", _rel(a[2], b[2]))
