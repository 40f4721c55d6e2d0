"""A/B of two builds of the library in ONE gpurun call (box-to-box step times differ by ~10 %, so only same-box
numbers compare): runs the training step with each library in alternating subprocesses.
usage: ab_lib.py <libA.so> <libB.so> [rounds]      (child mode: ab_lib.py --child <lib.so>)"""
import subprocess
import sys
import time

if sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, '/root/repo')
    from geomae_amd import _lib
    _lib.load(path=sys.argv[2])
    import geomae_amd
    from geomae_amd import synth
    from geomae_amd.configs import mae_sst_model
    from geomae_amd.train import Trainer
    dev = torch.device('cuda:0')
    torch.manual_seed(1234)
    cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
    model = geomae_amd.build_model(cfg).to(dev).train()
    tr = Trainer(model)
    B = 4
    pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
    step = lambda i: tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
    for i in range(8):
        step(i)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40):
            l, _ = step(i)
        torch.cuda.synchronize(); best = min(best, 1e3 * (time.perf_counter() - t0) / 40)
    print(f"{sys.argv[2].split('/')[-1]}: {best:.3f} ms/step (best of 3 x 40)  loss {float(sum(l.values())):.4f}", flush=True)
else:
    a, b = sys.argv[1], sys.argv[2]
    for r in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
        for lib in (a, b):
            subprocess.run([sys.executable, __file__, "--child", lib], check=False)
