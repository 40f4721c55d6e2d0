"""Per-rank compute schedule of a multi-GPU step, measured in one process: the step with an early-gradient hook (as the
trainer installs at world > 1; a no-op here) with and without the geometry-stream deferrals (the schedule before the
hook was allowed to ride on that stream).  usage: ab_early_hook.py"""
import subprocess, sys, time
if len(sys.argv) > 1:
    mode = sys.argv[1]
    sys.path.insert(0, "/root/repo")
    import torch, geomae_amd
    from geomae_amd import synth
    from geomae_amd.configs import mae_sst_model
    from geomae_amd.train import Trainer
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
    model = geomae_amd.build_model(cfg).to(dev).train()
    tr = Trainer(model)
    inner = model.train_step_explicit
    if mode != "plain":
        model.train_step_explicit = lambda p, on_early_grads=None, next_points=None: inner(p, on_early_grads=lambda: None, next_points=next_points)
    if mode == "hook_noside":
        bb = model.backbone
        orig = bb.losses_and_grads_explicit
        def patched(*a, **k):
            k["bufs"] = {kk: v for kk, v in k["bufs"].items() if kk != "side"}
            return orig(*a, **k)
        bb.losses_and_grads_explicit = patched
    B = 4
    pool = [[torch.as_tensor(synth.lidar_frame(10000 + i * B + b), device=dev) for b in range(B)] for i in range(4)]
    step = lambda i: tr.train_step(pool[i % 4], next_points=pool[(i + 1) % 4])
    for i in range(8): step(i)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): l, _ = step(i)
        torch.cuda.synchronize(); best = min(best, 1e3 * (time.perf_counter() - t0) / 40)
    print(f"{mode}: {best:.3f} ms/step  loss {float(sum(l.values())):.4f}", flush=True)
else:
    for r in range(2):
        for m in ("plain", "hook", "hook_noside"):
            subprocess.run([sys.executable, __file__, m], check=False)
