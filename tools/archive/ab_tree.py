"""A/B of two SOURCE TREES (python + library) in ONE gpurun call, alternating subprocesses: for changes that alter the
C ABI, where tools/ab_lib.py's library swap cannot apply.  Make the old tree inside the repo so that it travels:
  mkdir _ab_old && git archive HEAD | tar -x -C _ab_old && (cd _ab_old && python geomae_amd/csrc/build.py)
usage: ab_tree.py <treeA> <treeB> [rounds]"""
import subprocess, sys, time
if sys.argv[1] == "--child":
    tree = sys.argv[2]
    sys.path.insert(0, tree)
    import torch
    import geomae_amd
    assert geomae_amd.__file__.startswith(tree), geomae_amd.__file__
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
    print(f"{tree}: {best:.3f} ms/step (best of 3 x 40)  loss {float(sum(l.values())):.6f}", flush=True)
else:
    a, b = sys.argv[1], sys.argv[2]
    for r in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
        for t in (a, b):
            subprocess.run([sys.executable, __file__, "--child", t], check=False)
