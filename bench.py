"""GeoMAE-SST pre-training throughput on MI355X (BASELINE.json metric: pretrain frames/sec).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`
(one rank per GPU, RCCL).  A step = forward_train + backward + gradient all-reduce + clip + AdamW on
one batch of 4 synthetic nuScenes-like single-sweep frames per GPU (BASELINE config 2; weak scaling).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def cpu_baseline(seed=1000):
    """The oracle (CPU restatement of the reference path, kind='port') timed on the host cores on a
    bounded sample: ONE single-sweep frame, forward + backward (no optimizer)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import geomae_oracle as O
    from geomae_amd import synth
    cores = min(16, os.cpu_count() or 1)          # more threads only slow these small CPU ops down
    torch.set_num_threads(cores)
    cfg = O.mae_sst_cfg(6, 2)
    params = {k: v.requires_grad_(True) for k, v in O.make_params(7, 6, 2, perturb=False).items()}
    frame = synth.lidar_frame(seed)
    g = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    n = 0
    while True:
        losses, _ = O.forward_train(params, [frame], cfg, generator=g)
        sum(losses.values()).backward()
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 3:
            break
    return dict(value=round(n / dt, 4), unit="frames/s", cores=cores, kind="port",
                sample=f"{n} x (1 single-sweep frame, {frame.shape[0]} pts, fwd+bwd, torch-CPU fp32)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-gpu", type=int, default=4)
    ap.add_argument("--sweeps", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="instrument the dominant kernel's launches with HIP events in every N-th timed step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # GEOMAE_BENCH_SHARE_GPU=1 (test hook): every rank on cuda:0 with gloo collectives, to exercise the N > 1 code
    # path on a one-GPU box; the driver's multi-GPU runs use one GPU per rank and RCCL ("nccl")
    share = os.environ.get("GEOMAE_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
        os.environ.setdefault("GEOMAE_SIDE_STREAMS", "3")      # see geomae_amd.ops.side_streams
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"

    import geomae_amd
    from geomae_amd import _lib, ops, synth
    from geomae_amd.configs import mae_sst_model
    from geomae_amd.train import Trainer
    _lib.load()                                   # no fallback: fail here if the HIP library is missing

    torch.manual_seed(1234)                       # identical initial weights on every rank (as DDP broadcast)
    cfg = mae_sst_model()
    cfg["backbone"]["compute_dtype"] = args.dtype
    model = geomae_amd.build_model(cfg).to(dev).train()
    trainer = Trainer(model)
    B = args.frames_per_gpu
    pool = []
    for i in range(4):                            # 4 distinct batches per rank, cycled; resident in HBM
        pool.append([torch.as_tensor(synth.lidar_frame(10_000 * (rank + 1) + i * B + b, sweeps=args.sweeps), device=dev)
                     for b in range(B)])
    n_pts = float(np.mean([sum(p.shape[0] for p in batch) for batch in pool]))

    def step(i):
        # like a data loader that has batch i+1 ready: its voxelization / pillar sort is enqueued during step i
        # (Trainer.train_step next_points), so every timed step contains exactly one such stage -- the one of
        # the following batch -- and no step waits on its pillar-count readback
        return trainer.train_step(pool[i % len(pool)], next_points=pool[(i + 1) % len(pool)])

    for i in range(args.warmup):
        step(i)
    # (the stand-alone dw_kernel launches are not in the list: the explicit schedule defers them to the geometry stream,
    # geomae_flush_weight_grad, so the stack's event pair would bracket their recording, not their execution;
    # profiles/*kernel_stats.csv has their durations)
    TIMED = ("sst_ffn_bwd_kernel", "win_attn_bwd_kernel", "sst_ffn_fwd_kernel", "sst_qkv_bwd_kernel",
             "win_attn_fwd_kernel", "sst_qkv_fwd_kernel")
    DOMINANT = "sst_ffn_bwd_kernel"               # largest share in profiles/ (rocprofv3 --kernel-trace --stats)
    lib = _lib.load()
    import ctypes

    def profile_on(name, launches):
        ops.PROFILER = lib.geomae_profiler_create(ops.KERNEL_IDS[name], launches)
        assert ops.PROFILER

    def profile_off():
        buf = (ctypes.c_float * 4096)()
        n = lib.geomae_profiler_read(ctypes.c_void_p(ops.PROFILER), buf, 4096)
        lib.geomae_profiler_destroy(ctypes.c_void_p(ops.PROFILER))
        ops.PROFILER = None
        return [float(buf[i]) for i in range(n)]

    # HIP events on the launch stream around every launch of the dominant kernel, inside the timed region, in every
    # `--profile-every`-th timed step (default 4): two event records per launch cost ~4 us of queue time each, 0.14 ms
    # per step when every step is instrumented, and the timed region is what `value` is computed from
    profile_on(DOMINANT, min(4000, 20 * args.steps))
    prof_handle, ops.PROFILER = ops.PROFILER, None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.PROFILER = prof_handle if i % max(1, args.profile_every) == 0 else None
        losses, _ = step(args.warmup + i)
    ops.PROFILER = prof_handle
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    durations = {DOMINANT: profile_off()}
    for k in TIMED:                               # the other kernels: 3 extra (untimed) steps each
        if k != DOMINANT:
            profile_on(k, 64)
            for i in range(3):
                step(i)
            durations[k] = profile_off()
    loss_val = float(sum(v.detach() for v in losses.values()))
    assert np.isfinite(loss_val), "non-finite loss"

    if rank == 0:
        # Per-launch roofline of the hand-written layer kernels.  Durations: HIP events recorded on the launch
        # stream around every launch of the timed region.  Algorithmic FLOPs per launch (2 per MAC, DESIGN.md
        # section 3): projections 2*n*K*N over the tokens n of the layer; attention 2*{2,5}*16*H*sum_w n_w^2.
        with torch.no_grad():
            pts = pool[(args.warmup + args.steps - 1) % len(pool)]
            voxels, coors, _, _ = model.voxelize_all(pts)
            seg = ops.pillar_segment(coors, B, model.grid_size)
            vc = seg.voxel_coors[:seg.V]
            ids_keep, _, _, _ = ops.random_mask(seg, 1 - model.random_mask_ratio, 1)
            n_tok, sq = [], []
            for toks, layers in ((vc[ids_keep.long()], 12), (vc, 8)):
                for s_ in (0, 1):
                    L = ops.window_build(toks.contiguous(), B, model.backbone._wcfg, s_)
                    W = int(L.num_windows.item())
                    nw = (L.win_start[1:W + 1] - L.win_start[:W]).double()
                    sq += [float((nw * nw).sum())] * (layers // 2)
                    n_tok += [int(toks.shape[0])] * (layers // 2)
        n_sum, sq_sum = float(np.sum(n_tok)), float(np.sum(sq))          # over the 20 layers of one step
        # Fusion structure of a stack of L layers (12-layer encoder, two 4-layer decoders; csrc/sst_stack.hip):
        #   forward : F1(0) | attn | F3(l)+F1(l+1) ... | F3(L-1)             -> 17 of 20 ffn-forward launches carry an F1
        #   backward: B3(L-1) | battn | B1(l+1)+B3(l) [+ dW(l+1)] ... | B1(0) | dW(0)
        #                                                                     -> 17 of 20 ffn-backward launches carry B1 + dW
        # 3 stand-alone F1 / B1 / dW launches remain per step (one per stack).
        n_e, n_d = float(n_tok[0]), float(n_tok[-1])
        carried, alone = 11 * n_e + 6 * n_d, n_e + 2 * n_d
        flops_step = {"sst_ffn_bwd_kernel": 2 * 81920 * n_sum + 2 * (49152 + 131072) * carried,
                      "sst_ffn_fwd_kernel": 2 * 81920 * n_sum + 2 * 49152 * carried,
                      "sst_qkv_fwd_kernel": 2 * 49152 * alone, "sst_qkv_bwd_kernel": 2 * 49152 * alone,
                      "dw_kernel": 2 * 131072 * alone, "win_attn_fwd_kernel": 2 * 2 * 16 * 8 * sq_sum,
                      "win_attn_bwd_kernel": 2 * 5 * 16 * 8 * sq_sum}
        launches_step = {k: 20.0 for k in flops_step}
        launches_step.update({"dw_kernel": 3.0, "sst_qkv_fwd_kernel": 3.0, "sst_qkv_bwd_kernel": 3.0})
        report_name = {"sst_ffn_bwd_kernel": "sst_ffn_bwd_dw_kernel"}
        peak = 2500.0                                             # dense bf16 MFMA TFLOP/s (MI355X_MICROARCH.md)
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))
        kern = {}
        for k in TIMED:
            d = durations.get(k) or []
            if d:
                ms_step = float(np.sum(d)) / (len(d) / launches_step[k])
                ach = flops_step[k] / (ms_step * 1e-3) / 1e12
                tr_b = traffic.get(report_name.get(k, k))
                kern[k] = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 5), "traffic": tr_b,
                           "avg_launch_ms": round(float(np.mean(d)), 5), "launches_timed": len(d),
                           "ms_per_step": round(ms_step, 4)}
                if tr_b:        # the same launch against the HBM roof (8 TB/s): measured PMC bytes / measured duration
                    gbs = tr_b / (float(np.mean(d)) * 1e-3) / 1e9
                    kern[k]["hbm_view"] = {"achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                           "frac": round(gbs / 8000.0, 4)}
        dominant = DOMINANT
        out = {
            "metric": "pretrain frames/sec (nuScenes SST-GeoMAE)",
            "value": round(world * B * args.steps / elapsed, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs[{1 if args.sweeps == 1 else 2}]: nuScenes-like {args.sweeps}-sweep GeoMAE-SST pretrain (mae_sst model "
                                   f"6+2+2 blocks), {B} frames/GPU, ~{int(n_pts / B)} pts/frame, fwd+bwd+allreduce+clip+AdamW",
                       "frames_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}"},
            "loss": round(loss_val, 4),
            "roofline": dict(kernel=report_name.get(dominant, dominant), **kern[dominant]),
            "roofline_other_kernels": {report_name.get(k, k): v for k, v in kern.items() if k != dominant},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
