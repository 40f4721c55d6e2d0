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
    bounded sample: single-sweep frames one at a time, forward + backward (no optimizer), for ~12 s."""
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
        if (dt > 12.0 and n >= 3) or n >= 64:      # a bounded sample: ~12 s of CPU work
            break
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=round(n / dt, 4), unit="frames/s", cores=cores, kind="port", cpu_model=model,
                sample=f"{n} x (1 single-sweep frame, {frame.shape[0]} pts, fwd+bwd, torch-CPU fp32)")


WORKLOADS = {
    # name: (config index in BASELINE.json, sweeps, lidar_frame kwargs, model geometry kwargs)
    "nuscenes1": (1, 1, {}, {}),
    "nuscenes10": (2, 10, {}, {}),
    # Waymo-like geometry of configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:8-10 (no GeoMAE Waymo config
    # exists in the reference: synthesised): 64 beams, ~180 k points per frame
    "waymo": (3, 1, dict(beams=64, n_az=3100, pc_range=(-74.88, -74.88, -2.0, 74.88, 74.88, 4.0), elev=(-17.6, 2.4),
                         n_cyl=90, max_range=110.0),
              dict(voxel_size=(0.32, 0.32, 6), sub_voxel_size_low=(0.08, 0.08, 0.75), sub_voxel_size_med=(0.16, 0.16, 1.5),
                   point_cloud_range=(-74.88, -74.88, -2.0, 74.88, 74.88, 4.0), grid_size=(1, 468, 468))),
}


def hbm_kernels(model, pts, B, torch, ops):
    """HBM GB/s of the scatter-side kernels (north_star: 'rocprof HBM GB/s on the scatter'): each op timed alone with
    HIP events on the launch stream (20 repetitions), against its ALGORITHMIC bytes (SURVEY 8(d) / DESIGN section 3)."""
    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    out = {}
    with torch.no_grad():
        voxels, top, med, low = model.voxelize_all(pts)
        N = voxels.shape[0]
        boffs = torch.tensor([0] + list(np.cumsum([p.shape[0] for p in pts])), dtype=torch.int32, device=voxels.device)
        gz, gy, gx = model.grid_size
        cells = B * gz * gy * gx
        seg = ops.pillar_segment(top, B, model.grid_size)
        V = seg.V
        ve = model.voxel_encoder
        prepared = ve.prepare_points(voxels, seg)
        ik, im, token_row, counts = ops.random_mask(seg, 1 - model.random_mask_ratio, 1)
        M = int(im.numel())

        def vfe_fwd():
            ve.forward_explicit(voxels, seg, prepared=prepared)
        cases = {
            "voxelize_kernel": (lambda: ops.voxelize_batch3(voxels, boffs, B, model.voxel_size, model.sub_voxel_size_med,
                                                            model.sub_voxel_size_low, model.point_cloud_range),
                                N * (20 + 48), "N*(20 B read + 3*16 B written)"),
            "pillar_segment (hist+scan x3+place)": (lambda: ops.pillar_segment(top, B, model.grid_size),
                                                    N * 24 + cells * 8, "N*24 B + cells*8 B"),
            "vfe_prepare + vfe forward sweeps": (lambda: (ve.prepare_points(voxels, seg), vfe_fwd()),
                                                 N * 24 + V * 512, "N*(20+4) B + V*128*4 B (SURVEY 8(d))"),
            "geometry_targets (centroid/occ/normal)": (lambda: ops.geometry_targets(voxels, seg, med, low, model._tcfg, token_row,
                                                                                    counts, n_rows=M),
                                                       N * 56 + M * 1900, "N*56 B + M*1.9 KB"),
        }
        for name, (fn, nbytes, formula) in cases.items():
            ms = timed(fn)
            gbs = nbytes / (ms * 1e-3) / 1e9
            out[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 5),
                         "avg_ms": round(ms, 5), "algorithmic_bytes": int(nbytes), "bytes_formula": formula}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-gpu", type=int, default=4)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--sweeps", type=int, default=1, help="(older spelling) 1 = --workload nuscenes1, 10 = nuscenes10")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-engine", action="store_true", help="Python explicit schedule instead of the C step engine (A/B)")
    ap.add_argument("--profile-every", type=int, default=0, help="(ignored: the timed region is no longer instrumented)")
    args = ap.parse_args()
    workload = args.workload or ("nuscenes10" if args.sweeps == 10 else "nuscenes1")
    cfg_index, sweeps, frame_kw, geom_kw = WORKLOADS[workload]
    if args.workload is None and args.sweeps not in (1, 10):
        sweeps = args.sweeps

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
    # GEOMAE_FORCE_EXCHANGE=1 at N = 1: the world > 1 schedule (engine hooks, SyncBN exchanges, early gradient-segment
    # all-reduces, optimizer as a second call) over RCCL with ONE rank -- what the stream-ordered collective path costs
    # on top of the plain N = 1 step (geomae_amd.train.exchange_mode)
    forced = world == 1 and os.environ.get("GEOMAE_FORCE_EXCHANGE") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # A mis-launched run must not print a plausible line (reference launcher: tools/dist_train.sh:8-9, one rank per GPU):
    # --gpus N means N ranks in THIS process group, each on a GPU of its own.
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 with "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}`")
    if world > 1:
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        mine = [rank, int(torch.cuda.current_device()), str(torch.cuda.get_device_properties(dev).uuid)
                if hasattr(torch.cuda.get_device_properties(dev), "uuid") else str(local_rank)]
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        if not share and len({(g_[1], g_[2]) for g_ in seen}) != world:
            raise SystemExit(f"bench.py: two ranks share a GPU under RCCL (rank, device, uuid): {seen}")

    import geomae_amd
    from geomae_amd import _lib, ops, synth
    from geomae_amd.configs import mae_sst_model
    from geomae_amd.train import Trainer
    _lib.load()                                   # no fallback: fail here if the HIP library is missing

    torch.manual_seed(1234)                       # identical initial weights on every rank (as DDP broadcast)
    cfg = mae_sst_model(**geom_kw)
    if geom_kw:
        cfg["backbone"]["output_shape"] = list(geom_kw["grid_size"][1:])
    cfg["backbone"]["compute_dtype"] = args.dtype
    model = geomae_amd.build_model(cfg).to(dev).train()
    trainer = Trainer(model)
    if args.no_engine:
        trainer.use_engine = False
    B = args.frames_per_gpu
    pool = []
    for i in range(4):                            # 4 distinct batches per rank, cycled; resident in HBM
        pool.append([torch.as_tensor(synth.lidar_frame(10_000 * (rank + 1) + i * B + b, sweeps=sweeps, **frame_kw), device=dev)
                     for b in range(B)])
    n_pts = float(np.mean([sum(p.shape[0] for p in batch) for batch in pool]))

    def step(i):
        # like a data loader that has batch i+1 ready: its voxelization / pillar sort is enqueued during step i
        # (Trainer.train_step next_points), so every timed step contains exactly one such stage -- the one of
        # the following batch -- and no step waits on its pillar-count readback
        return trainer.train_step(pool[i % len(pool)], next_points=pool[(i + 1) % len(pool)])

    for i in range(args.warmup):
        step(i)
    # The layer kernels, under the names of the rocprofv3 table.  The ffn launches come in two kernels each
    # (sst_ffn_bwd_kernel / sst_ffn_bwd_dw_kernel: without / with a weight-gradient contraction riding along;
    # sst_ffn_fwd_kernel / sst_ffn_fwd_pair_kernel); sst_layer_fwd_kernel is the one-launch layer forward (small token
    # sets); dw_kernel's launches are deferred to the geometry stream and timed THERE (thread profiler, csrc/sst_layer.hip).
    TIMED = ("sst_ffn_bwd_kernel", "sst_ffn_bwd_dw_kernel", "win_attn_bwd_kernel", "sst_ffn_fwd_kernel",
             "sst_ffn_fwd_pair_kernel", "sst_qkv_bwd_kernel", "win_attn_fwd_kernel", "sst_qkv_fwd_kernel",
             "sst_layer_fwd_kernel", "sst_layer_bwd_kernel", "dw_kernel")
    # The kernel instrumented INSIDE the timed region is the TIMED kernel with the largest share of the committed
    # rocprofv3 --kernel-trace --stats table of this workload (profiles/rNN_<workload>_kernel_stats.csv, newest round); the
    # others are timed in extra steps behind it, and `roofline` reports whichever turned out largest by measured time per
    # step (with `dominant_check` saying whether table and measurement agree).
    def committed_table():
        import csv, glob, re
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{workload}_kernel_stats.csv")))
        if not files:
            return None, {}
        share = {}
        for r in csv.DictReader(open(files[-1])):
            name = re.sub(r"<.*", "", r["Name"].split("(")[0].replace("geomae::", "").replace("void ", "")).strip()
            name = {"dw_layer_kernel": "dw_kernel"}.get(name, name)      # (one profiler id: the contraction in both forms)
            share[name] = share.get(name, 0.0) + float(r["Percentage"])
        return os.path.relpath(files[-1], ROOT), share
    table_path, table_share = committed_table()
    DOMINANT = max(TIMED, key=lambda k: table_share.get(k, 0.0)) if table_share else "sst_ffn_bwd_kernel"
    lib = _lib.load()
    import ctypes

    def set_profiler(handle):
        ops.PROFILER = handle
        if trainer.engine is not None:            # (looked up at every use: None before the first step / with --no-engine)
            trainer.engine.set_profiler(handle)

    def profile_on(name, launches):
        h = lib.geomae_profiler_create(ops.KERNEL_IDS[name], launches)
        assert h
        return h

    def profile_off(h):
        buf = (ctypes.c_float * 4096)()
        n = lib.geomae_profiler_read(ctypes.c_void_p(h), buf, 4096)
        set_profiler(None)
        lib.geomae_profiler_destroy(ctypes.c_void_p(h))
        return [float(buf[i]) for i in range(n)]

    # The timed region carries NO instrumentation: no per-launch profiler handle, no per-step events.  Per-launch
    # durations (HIP events on each kernel's launch stream), the per-step distribution and the phase times all come
    # from EXTRA steps behind it.
    eng = trainer.get_engine()                    # (exists before the first step too: --warmup 0)
    h0 = eng.host_times() if eng is not None else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses, _ = step(args.warmup + i)
    t_enq = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    h1 = eng.host_times() if eng is not None else None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # per-step distribution: one event per step on the main stream (~4 us of queue time each), 16 extra steps
    n_dist = 16
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(n_dist + 1)]
    step_events[0].record()
    for i in range(n_dist):
        step(args.warmup + args.steps + i)
        step_events[i + 1].record()
    torch.cuda.synchronize()
    per_step = np.array([step_events[i].elapsed_time(step_events[i + 1]) for i in range(n_dist)])
    durations = {}
    steps_timed = {k: 3 for k in TIMED}
    for k in TIMED:                               # every kernel alike: 3 extra (untimed) steps each
        h = profile_on(k, 64 * 3)
        set_profiler(h)
        for i in range(3):
            step(i)
        durations[k] = profile_off(h)
    # what an event pair with NOTHING between its records measures on this stream (the two records' own queue time): the
    # per-launch durations above contain it, rocprofv3's kernel durations do not
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a_, b_ in pairs:
        a_.record()
        b_.record()
    torch.cuda.synchronize()
    event_pair_ms = float(np.median([a_.elapsed_time(b_) for a_, b_ in pairs]))
    phases, exposed = None, None
    if eng is not None:                           # where the main stream's time goes: 3 instrumented (untimed) steps
        eng.set_phase_timing(True)
        measure_comm = world > 1 or forced
        if measure_comm:                          # what the main stream waits for: in-line SyncBN all-reduces, the last segment
            eng.comm_events, trainer.tail_comm_events = [], []
        acc = {}
        for i in range(3):
            step(i)
            for k, v in eng.phase_times().items():
                acc[k] = acc.get(k, 0.0) + v / 3
        eng.set_phase_timing(False)
        phases = {k: round(v, 4) for k, v in acc.items()}
        if measure_comm:
            torch.cuda.synchronize()
            names = {0: "bn_fwd0", 1: "bn_fwd1", 2: "bn_bwd1", 3: "bn_bwd0"}
            per = {}
            for what, e0, e1 in eng.comm_events:
                per[names[what]] = per.get(names[what], 0.0) + e0.elapsed_time(e1) / 3
            tail_ms = sum(e0.elapsed_time(e1) for e0, e1 in trainer.tail_comm_events) / 3
            exposed = {"syncbn_inline_ms_per_step": {k: round(v, 4) for k, v in per.items()},
                       "syncbn_inline_total_ms_per_step": round(sum(per.values()), 4),
                       "last_gradient_segment_wait_ms_per_step": round(tail_ms, 4),
                       "note": "HIP event pairs on the main stream around each in-line SyncBN all-reduce and around the "
                               "final gradient-segment all-reduce + wait (the two early segments and the feature-moment "
                               "exchange run on side streams beside compute); 3 untimed steps"}
            eng.comm_events = trainer.tail_comm_events = None
    rank_devices = None
    if world > 1 or forced:                       # every rank's device (index, name), gathered on the run's own process group
        mine = [rank, int(torch.cuda.current_device()), torch.cuda.get_device_name(dev), int(local_rank)]
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        rank_devices = [{"rank": g_[0], "cuda_device": g_[1], "name": g_[2], "local_rank": g_[3]} for g_ in gathered]
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
        #   forward : F1(0) | attn | F3(l)+F1(l+1) ... | F3(L-1)      -> L-1 of a stack's L ffn-forward launches carry an F1
        #   backward: B3(L-1) | battn | B1(l+1)+B3(l) [+ dW(l+1)] ... | B1(0) | dW(0)
        # Round 5 (engine default): NO launch carries a contraction -- 20 plain sst_ffn_bwd_kernel launches per step (17 with a
        # B1 head); every layer's contraction runs in the layer-form launches on the geometry stream (profiler id "dw_kernel":
        # 2 decoder launches of 4 layers, 3 encoder launches of 4 layers, + the old-form heads and VFE launches).
        # GEOMAE_ENC_DW_DEFER=0: the encoder's launches carry dW(l+1) again (sst_ffn_bwd_dw_kernel, 11 per step);
        # GEOMAE_DW_DEFER_ALL=0 / the python step driver: the decoders' 6 non-top launches too.
        # 3 stand-alone F1 / B1 launches per step (one per stack).
        n_e, n_d = float(n_tok[0]), float(n_tok[-1])
        sq_e, sq_d = float(np.sum(sq[:12])), float(np.sum(sq[12:]))   # sum over layers of sum_w n_w^2 (encoder / one decoder)
        F3, F1, DW = 2 * 81920.0, 2 * 49152.0, 2 * 131072.0           # FLOPs per token: ffn (3 GEMMs) / qkv / contractions
        lps = lambda k: (len(durations.get(k) or []) / steps_timed[k]) if durations.get(k) else 0.0     # launches per step
        fused_enc = round(lps("sst_layer_fwd_kernel")) == 12          # the encoder's forward as one launch per layer
        dec_deferred = round(lps("sst_ffn_bwd_dw_kernel")) == 11
        # round 5: the encoder's contractions leave its backward launches too (csrc/engine.hip GEOMAE_ENC_DW_DEFER): no
        # sst_ffn_bwd_dw_kernel launch at all, 20 plain ffn-backward launches (17 of them with a B1 head), and every layer's
        # contraction in the layer-form launches on the geometry stream (csrc/dw_device.h: <= 4 layers per launch)
        all_deferred = round(lps("sst_ffn_bwd_dw_kernel")) == 0 and round(lps("sst_ffn_bwd_kernel")) in (20, 8)
        # ... and the encoder's backward as ONE launch per layer (csrc/sst_fused.hip sst_layer_bwd_kernel, 12 per step): the
        # ffn-backward / attention-backward / stand-alone in-projection-backward launches left are the decoders' (8 / 8 / 2)
        fused_bwd = round(lps("sst_layer_bwd_kernel")) == 12
        M_rows = float(n_d - n_e)
        ATT_B = 2 * 5 * 16 * 8                                          # attention backward FLOPs per (query, key) pair of a window
        flops_step = {"sst_layer_bwd_kernel": (F3 + F1) * 12 * n_e + ATT_B * sq_e,
                      "sst_ffn_bwd_kernel": (F3 * 8 * n_d + F1 * 6 * n_d) if fused_bwd else
                                            (F3 * (12 * n_e + 8 * n_d) + F1 * (11 * n_e + 6 * n_d)) if all_deferred else
                                            (F3 * (n_e + 8 * n_d) + F1 * 6 * n_d if dec_deferred else F3 * (n_e + 2 * n_d)),
                      "sst_ffn_bwd_dw_kernel": (F3 + F1 + DW) * (11 * n_e + (0 if dec_deferred else 6 * n_d)),
                      "sst_ffn_fwd_kernel": F3 * 8 * n_d + F1 * 6 * n_d,
                      "sst_ffn_fwd_pair_kernel": F3 * 12 * n_e + F1 * 11 * n_e,
                      "sst_qkv_fwd_kernel": F1 * ((0 if fused_enc else n_e) + 2 * n_d),
                      "sst_qkv_bwd_kernel": F1 * ((0 if fused_bwd else n_e) + 2 * n_d),
                      "win_attn_fwd_kernel": 2 * 2 * 16 * 8 * (sq_sum - (sq_e if fused_enc else 0.0)),
                      "win_attn_bwd_kernel": ATT_B * (sq_sum - (sq_e if fused_bwd else 0.0)),
                      "sst_layer_fwd_kernel": (F3 + F1) * 12 * n_e + 2 * 2 * 16 * 8 * sq_e,
                      # every stand-alone contraction of the step: the decoders' 8 layers + the encoder's first layer (its
                      # other 11 ride in sst_ffn_bwd_dw_kernel), the six heads (800 x 128 per masked row), VFE layer 1
                      "dw_kernel": DW * ((8 * n_d if (dec_deferred or all_deferred) else 2 * n_d) + (12 * n_e if all_deferred else n_e))
                                   + 2 * 800 * 128 * M_rows + 2 * 128 * 128 * n_pts}
        expect = {"sst_layer_bwd_kernel": 12.0,
                  "sst_ffn_bwd_kernel": 8.0 if fused_bwd else 20.0 if all_deferred else (9.0 if dec_deferred else 3.0),
                  "sst_ffn_bwd_dw_kernel": 11.0 if dec_deferred else 17.0,
                  "sst_ffn_fwd_kernel": 8.0, "sst_ffn_fwd_pair_kernel": 12.0, "sst_qkv_fwd_kernel": 2.0 if fused_enc else 3.0,
                  "sst_qkv_bwd_kernel": 2.0 if fused_bwd else 3.0, "win_attn_fwd_kernel": 8.0 if fused_enc else 20.0,
                  "win_attn_bwd_kernel": 8.0 if fused_bwd else 20.0,
                  "sst_layer_fwd_kernel": 12.0, "dw_kernel": None}
        launches_step = {k: lps(k) for k in flops_step}
        report_name = {}
        peak = 2500.0                                             # dense bf16 MFMA TFLOP/s (MI355X_MICROARCH.md)
        # HBM bytes per launch are NOT measured in this run: they come from the stored PMC passes of tools/pmc.sh
        # (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc runs) and only apply to the workload they were taken on
        traffic, traffic_src = {}, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{workload}_pmc_traffic.json")), reverse=True)
        if workload == "nuscenes1":
            cands += [os.path.join(ROOT, "profiles", c) for c in ("r02_pmc_traffic.json", "r01_pmc_traffic.json")]
        for tpath in cands:
            if B == 4 and os.path.exists(tpath):
                traffic, traffic_src = json.load(open(tpath)), os.path.relpath(tpath, ROOT)
                break
        # Algorithmic HBM bytes per step of the same kernels (per token and layer): F3 reads x (fp32 512 B)
        # + attn (bf16 256) and writes xhat1 256 + hp 512 + xhat2 256 + z 512; a riding F1 writes x+pos 256 + qkv 768; B3 moves
        # 4.45 KB (reads dz 512, xhat1/xhat2/hp/attn 1280, writes dv/dhp/h/du/dx_res/dattn 2304, + LayerNorm rows), a B1 head
        # 1.3 KB; a contraction reads 8 tasks x 2 operands x 256 B; attention forward reads qkv 768 and writes attn 256 + lse 32,
        # its backward reads qkv + attn + dattn + lse and writes dqkv 768.
        # The contraction: 3.25 KB per token and layer = 13 x 256-byte operand rows (the four jobs of
        # csrc/dw_device.h read [dq | dk] 512 + xp 256, dhp 512 + xhat1 256, dv 256 + h 512, dv_ 256 + x 256 + du 256 + attn 256;
        # task pairs that share an operand read it once -- the round-4 task form read 4 KB).
        F3B, F1B, B3B, B1B, DWB, AFB, ABB = 2304.0, 1024.0, 4450.0, 1300.0, 3328.0, 1056.0, 2080.0
        # the one-launch backward: reads dz 512 + xhat1/xhat2/hp/attn/qkv 1792 + lse/rstd 40, writes dx 512 + dv/dhp/h/du/dqkv 2304
        bytes_step = {"sst_layer_bwd_kernel": (512 + 1792 + 40 + 512 + 2304) * 12 * n_e,
                      "sst_ffn_bwd_kernel": (B3B * 8 * n_d + B1B * 6 * n_d) if fused_bwd else
                                            (B3B * (12 * n_e + 8 * n_d) + B1B * (11 * n_e + 6 * n_d)) if all_deferred else
                                            (B3B * (n_e + 8 * n_d) + B1B * 6 * n_d if dec_deferred else B3B * (n_e + 2 * n_d)),
                      "sst_ffn_bwd_dw_kernel": (B3B + B1B + DWB) * (11 * n_e + (0 if dec_deferred else 6 * n_d)),
                      "sst_ffn_fwd_kernel": F3B * 8 * n_d + F1B * 6 * n_d,
                      "sst_ffn_fwd_pair_kernel": F3B * 12 * n_e + F1B * 11 * n_e,
                      "sst_qkv_fwd_kernel": (512 + F1B) * ((0 if fused_enc else n_e) + 2 * n_d),
                      "sst_qkv_bwd_kernel": (B1B + 512) * ((0 if fused_bwd else n_e) + 2 * n_d),
                      "win_attn_fwd_kernel": AFB * (8 * n_d + (0 if fused_enc else 12 * n_e)),
                      "win_attn_bwd_kernel": ABB * ((0 if fused_bwd else 12 * n_e) + 8 * n_d),
                      "sst_layer_fwd_kernel": (512 + F3B + F1B + 256 + 32) * 12 * n_e,
                      # (3328 B per token-layer; + the heads' rows and the VFE layer-1 operands)
                      "dw_kernel": DWB * ((8 * n_d if (dec_deferred or all_deferred) else 2 * n_d) + (12 * n_e if all_deferred else n_e))
                                   + 2 * (800 + 128) * M_rows + 2 * 2 * 128 * n_pts}
        kern = {}
        for k in TIMED:
            d = durations.get(k) or []
            if d and (expect[k] is None or abs(launches_step[k] - expect[k]) < 0.01):      # (another launch structure: no FLOP model here)
                ms_step = float(np.sum(d)) / (len(d) / launches_step[k])
                net_launch = max(float(np.mean(d)) - event_pair_ms, 1e-4)            # less the empty event pair: ~ rocprofv3's AverageNs
                net_step = net_launch * launches_step[k]
                ach = flops_step[k] / (net_step * 1e-3) / 1e12
                gbs_alg = bytes_step[k] / (net_step * 1e-3) / 1e9
                ai = flops_step[k] / bytes_step[k]                                   # FLOP per algorithmic byte
                roof_tflops = min(peak, ai * 8.0)                                    # min(MFMA peak, AI x 8 TB/s)
                tr_b = traffic.get(report_name.get(k, k))
                common = {"arithmetic_intensity_flop_per_byte": round(ai, 1), "roof_at_this_intensity_tflops": round(roof_tflops, 1),
                          "frac_of_own_roof": round(ach / roof_tflops, 5),
                          "traffic": tr_b, "traffic_source": traffic_src if tr_b else None,
                          "avg_launch_ms": round(float(np.mean(d)), 5), "launches_timed": len(d),
                          "launches_per_step": round(launches_step[k], 2), "ms_per_step": round(net_step, 4),
                          "ms_per_step_with_event_pairs": round(ms_step, 4), "timed_in": "3 extra steps behind the timed region",
                          "avg_launch_ms_less_event_pair": round(net_launch, 5), "empty_event_pair_ms": round(event_pair_ms, 5),
                          "algorithmic_flop_per_launch": int(flops_step[k] / launches_step[k]),
                          "algorithmic_bytes_per_launch": int(bytes_step[k] / launches_step[k])}
                mfma_view = {"achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 5)}
                hbm_alg = {"achieved": round(gbs_alg, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs_alg / 8000.0, 5)}
                if ai * 8.0 < peak:      # the roof this kernel's arithmetic intensity puts it under is HBM
                    kern[k] = dict(bound="hbm", **hbm_alg, **common, mfma_view=mfma_view)
                else:
                    kern[k] = dict(bound="mfma", **mfma_view, **common, hbm_algorithmic_view=hbm_alg)
                if tr_b:        # the same launch against the HBM roof with the stored PMC bytes / measured duration
                    gbs = tr_b / (net_launch * 1e-3) / 1e9
                    kern[k]["hbm_view"] = {"achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                           "frac": round(gbs / 8000.0, 4),
                                           "traffic_over_algorithmic": round(tr_b / (bytes_step[k] / launches_step[k]), 3)}
        # The contraction launches are throttled ON PURPOSE (csrc/sst_layer.hip launch_dw_layers: 80 workgroups of one per CU for
        # token sets up to 32768, 96 above -- they share the chip with the backward of the next stack): what that many CUs can
        # stream at all is measured by tools/lds_dma_bench.hip (profiles/r05_microbench_lds_dma.txt: 512-thread workgroups moving
        # 24-KB slabs global -> LDS and nothing else), so the launch is also priced against THAT ceiling.
        if "dw_kernel" in kern:
            tl = (8 * n_d if (dec_deferred or all_deferred) else 2 * n_d) + (12 * n_e if all_deferred else n_e)
            kern["dw_kernel"]["bytes_formula"] = (
                f"(token-layers {int(tl)} x 3328 B [13 operand rows of 256 B, DESIGN.md section 3] + masked rows {int(M_rows)} x 1856 B "
                f"[heads: 800 + 128 bf16] + points {int(n_pts)} x 512 B [VFE layer 1: dy1, g]) / {launches_step['dw_kernel']:.2f} launches per step"
                " / avg_launch_ms_less_event_pair")
            # the same kernel ALONE on the chip (tools/dw_bench.hip, committed output): what the throttled in-step figure is not
            alone_path = os.path.join(ROOT, "profiles", "r06_microbench_dw_alone.txt")
            if os.path.exists(alone_path):
                import re as _re
                best = None
                for line in open(alone_path):
                    m_ = _re.search(r"n =\s*(\d+), (\d+) layer\(s\) per launch, G =\s*(\d+) \(\s*(\d+) workgroups\): ([0-9.]+) us with its reduction "
                                    r"\(([0-9.]+) without\) = ([0-9.]+) us per layer", line)
                    # (the decoder-size run of the file: the largest n, then the fastest launch shape)
                    if m_ and (best is None or (int(m_.group(1)), -float(m_.group(7))) > (best[1], -best[0])):
                        best = (float(m_.group(7)), int(m_.group(1)), int(m_.group(2)), int(m_.group(4)))
                if best:
                    us_layer, n_al, layers_al, wg_al = best
                    gbs_al = n_al * 3328.0 / (us_layer * 1e-6) / 1e9
                    kern["dw_kernel"]["alone"] = {"achieved": round(gbs_al, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs_al / 8000.0, 4),
                                                  "us_per_layer": us_layer, "tokens": n_al, "layers_per_launch": layers_al, "workgroups": wg_al,
                                                  "bytes_formula": "tokens x 3328 B / us_per_layer", "source": "profiles/r06_microbench_dw_alone.txt (tools/dw_bench.hip)"}
            ceil_path = os.path.join(ROOT, "profiles", "r05_microbench_lds_dma.txt")
            budget = 80 if n_d <= 32768 else 96
            ceiling = None
            if os.path.exists(ceil_path):
                import re as _re
                pts = {}
                for line in open(ceil_path):
                    m_ = _re.match(r"LDS-direct, ring 4\s+(\d+) workgroups:\s+([0-9.]+) GB/s", line)
                    if m_:
                        pts[int(m_.group(1))] = float(m_.group(2))
                xs = sorted(pts)
                if xs and xs[0] <= budget <= xs[-1]:
                    ceiling = float(np.interp(budget, xs, [pts[x] for x in xs]))
            moved = kern["dw_kernel"].get("hbm_view", {}).get("achieved")
            kern["dw_kernel"]["throttle"] = {
                "workgroups_per_launch": budget, "one_workgroup_per_cu": True,
                "stream_ceiling_at_that_many_workgroups_gbs": round(ceiling, 1) if ceiling else None,
                "ceiling_source": "profiles/r05_microbench_lds_dma.txt (tools/lds_dma_bench.hip, alone on the chip)" if ceiling else None,
                "frac_of_that_ceiling_pmc_bytes": round(moved / ceiling, 4) if (ceiling and moved) else None,
                "frac_of_that_ceiling_algorithmic_bytes": round(kern["dw_kernel"]["achieved"] / ceiling, 4)
                if (ceiling and kern["dw_kernel"].get("bound") == "hbm") else None,
                "note": "in the step the launches share HBM with the main stream's kernels; alone on the chip the same kernel moves "
                        "5.2-5.5 TB/s on 192-256 workgroups (tools/dw_bench.hip)"}
        # the dominant kernel: largest measured time per step LESS the empty event pairs (a kernel with many short launches
        # is not promoted by the instrumentation's own queue time) -- what the rocprofv3 table ranks by
        largest = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        dominant = largest
        dominant_check = {"committed_table": table_path, "largest_share_in_table": DOMINANT,
                          "largest_measured_ms_per_step": largest, "agree": largest == DOMINANT}
        # ---- the whole step against both roofs: algorithmic FLOPs of one step (2 per MAC; forward = VFE + 20 SST layers +
        # heads, a training step = 3 x forward: SURVEY 8(d)) and the stored PMC bytes of one step, each / ms_per_step / peak
        fwd_flops = (2.0 * n_pts * (11 * 64 + 128 * 128) + (F3 + F1) * (12 * n_e + 8 * n_d) + 2 * 2 * 16 * 8 * sq_sum
                     + 2.0 * M_rows * 128 * 800)
        step_ms = elapsed / args.steps * 1e3
        step_roofline = {"algorithmic_flop_per_step": int(3 * fwd_flops),
                         "mfma": {"achieved": round(3 * fwd_flops / (step_ms * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                  "frac": round(3 * fwd_flops / (step_ms * 1e-3) / 1e12 / peak, 5)}}
        if traffic.get("_bytes_per_step"):
            gbs = traffic["_bytes_per_step"] / (step_ms * 1e-3) / 1e9
            step_roofline["hbm"] = {"bytes_per_step": int(traffic["_bytes_per_step"]), "source": traffic_src, "achieved": round(gbs, 1),
                                    "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}
            # what a kernel that only copies gets on this part (tools/rw_mix_bench.hip: read + write mixes 3 : 1 ... 1 : 3)
            mix_path = os.path.join(ROOT, "profiles", "r05_microbench_rw_mix.txt")
            if os.path.exists(mix_path):
                import re as _re
                mix = [float(m_.group(1)) for m_ in (_re.search(r"workgroups:\s+([0-9.]+) GB/s", l_) for l_ in open(mix_path)
                                                     if l_.startswith("read ") and "write" in l_.split("workgroups")[0] and "only" not in l_) if m_]
                if mix:
                    step_roofline["hbm"]["copy_kernel_read_write_mix_gbs"] = [round(min(mix), 1), round(max(mix), 1)]
                    step_roofline["hbm"]["frac_of_copy_kernel_mean"] = round(gbs / (sum(mix) / len(mix)), 4)
                    step_roofline["hbm"]["copy_source"] = "profiles/r05_microbench_rw_mix.txt"
        # ---- north_star's "per bucket" MFMA utilisation.  This build has no length buckets: windows are packed into bundles
        # and the attention kernels issue 16 x 16 score tiles over the key-tile range of each query tile.  Density = useful
        # score entries (sum_w n_w^2) / (256 x tiles issued): what fraction of an issued attention MFMA's rows x columns
        # is a real (query, key) pair, per token set and shift, by bundle size class.
        def tile_density(L, which):
            NBn = int((L.num_fbundles if which == "layer" else L.num_bundles).item())
            bt = (L.fbun_tok if which == "layer" else L.bun_tok)[:NBn + 1].cpu().numpy().astype(np.int64)
            W = int(L.num_windows.item())
            ws = L.win_start[:W + 1].cpu().numpy().astype(np.int64)
            res = {}
            for lo_, hi_ in ((1, 32), (33, 64), (65, 96), (97, 144)):
                useful = issued = nb_ = 0
                for b_ in range(NBn):
                    s0_, s1_ = bt[b_], bt[b_ + 1]
                    T_ = s1_ - s0_
                    if not (lo_ <= T_ <= hi_):
                        continue
                    nb_ += 1
                    w0, w1 = np.searchsorted(ws, s0_), np.searchsorted(ws, s1_)
                    sizes = np.diff(ws[w0:w1 + 1])
                    useful += int((sizes * sizes).sum())
                    nt_ = (T_ + 15) // 16
                    if which == "layer" and nt_ <= 4:
                        issued += nt_ * nt_                          # the exact-size bodies compute every tile pair
                    else:
                        for it_ in range(nt_):                       # key-tile range of the windows touching query tile it_
                            first, last = s0_ + 16 * it_, min(s0_ + 16 * it_ + 15, s1_ - 1)
                            wf, wl_ = np.searchsorted(ws, first, side="right") - 1, np.searchsorted(ws, last, side="right") - 1
                            issued += (ws[wl_ + 1] - 1 - s0_) // 16 - (ws[wf] - s0_) // 16 + 1
                if nb_:
                    res[f"{lo_}-{hi_} tokens"] = {"bundles": nb_, "useful_entries": useful, "tiles_issued": int(issued),
                                                  "density": round(useful / (256.0 * issued), 4)}
            return res
        density = {}
        with torch.no_grad():
            for tag, toks in (("encoder", vc[ids_keep.long()]), ("decoder", vc)):
                for s_, L in enumerate(ops.window_build_batch([(toks.contiguous(), 0), (toks.contiguous(), 1)], B, model.backbone._wcfg)):
                    density[f"{tag} shift {s_} (attention kernels)"] = tile_density(L, "attn")
                    if tag == "encoder" and fused_enc:
                        density[f"{tag} shift {s_} (one-launch layer kernel)"] = tile_density(L, "layer")
        out = {
            "metric": "pretrain frames/sec (nuScenes SST-GeoMAE)",
            "value": round(world * B * args.steps / elapsed, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs[{cfg_index}]: {workload} ({sweeps}-sweep) GeoMAE-SST pretrain (mae_sst model "
                                   f"6+2+2 blocks), {B} frames/GPU, ~{int(n_pts / B)} pts/frame, fwd+bwd+allreduce+clip+AdamW",
                       "frames_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}",
                       "step_driver": "python-explicit" if eng is None else "geomae_pretrain_step (C engine)",
                       "exchange": "forced at world size 1 (RCCL, one rank)" if forced else
                                   (("rccl" if dist.get_backend() == "nccl" else dist.get_backend() + " (GEOMAE_BENCH_SHARE_GPU test hook)")
                                    if world > 1 else "none")},
            "loss": round(loss_val, 4),
            # how the step's side streams were chosen (geomae_amd.ops._pick_side_streams: measured queue sharing)
            "stream_probe": ops.STREAM_PROBE.get((dev.type, dev.index)),
            # GPU-side time between the ends of consecutive steps on the main stream (HIP events), rank 0
            "step_ms": {"p50": round(float(np.percentile(per_step, 50)), 4), "p90": round(float(np.percentile(per_step, 90)), 4),
                        "max": round(float(per_step.max()), 4), "min": round(float(per_step.min()), 4)},
            # host wall time to ENQUEUE the timed steps (everything but the final synchronize), per step; with the
            # engine also the share of it spent blocked on the pillar-count readback (= how far the host runs ahead)
            "host_ms_per_step": {"enqueue_loop": round(1e3 * t_enq / args.steps, 4)},
            "roofline": dict(kernel=report_name.get(dominant, dominant), **kern[dominant]) if dominant else None,
            "dominant_check": dominant_check,
            "roofline_other_kernels": {report_name.get(k, k): v for k, v in kern.items() if k != dominant},
            "step_roofline": step_roofline,
            "attention_tile_density": density,
        }
        if h0 is not None:
            busy = (h1[0] - h0[0]) - (h1[1] - h0[1])
            out["host_ms_per_step"].update(inside_engine=round(1e3 * (h1[0] - h0[0]) / args.steps, 4),
                                           blocked_on_count_readback=round(1e3 * (h1[1] - h0[1]) / args.steps, 4),
                                           busy=round(1e3 * (t_enq - (h1[1] - h0[1])) / args.steps, 4),
                                           engine_busy=round(1e3 * busy / args.steps, 4))
        # north_star: "MFMA utilisation on the bucketed attention".  Not measurable from inside this process (PMC needs
        # rocprofv3): the stored pass of tools/archive/r2_profile.sh for this workload, labelled as such
        ucands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{workload}_mfma_util.json")), reverse=True)
        if workload == "nuscenes1":
            ucands.append(os.path.join(ROOT, "profiles", "r02_mfma_util.json"))
        for upath in ucands:
            if B == 4 and os.path.exists(upath):
                u = json.load(open(upath))
                out["mfma_busy_stored"] = {"source": f"{os.path.relpath(upath, ROOT)} (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))",
                                           **{k: u[k]["mfma_busy_frac"] for k in ("win_attn_fwd_kernel", "win_attn_bwd_kernel",
                                                                                  "sst_layer_fwd_kernel", "sst_layer_bwd_kernel", "dw_layer_kernel", "sst_ffn_fwd_kernel",
                                                                                  "sst_ffn_fwd_pair_kernel", "sst_ffn_bwd_kernel",
                                                                                  "sst_ffn_bwd_dw_kernel", "dw_kernel",
                                                                                  "vfe_layer1_kernel") if k in u}}
                break
        if world > 1 or forced:      # the N > 1 line validates itself: what the process group says it is running on
            out["distributed"] = {"world_size_reported": dist.get_world_size(), "backend": dist.get_backend(),
                                  "rank_devices": rank_devices, "exposed_communication": exposed}
        if phases is not None:
            out["main_stream_phase_ms"] = phases
            out["main_stream_phase_sum_ms"] = round(float(sum(phases.values())), 4)
        if world == 1:          # (stand-alone op calls: at world > 1 the VFE's SyncBN collectives need every rank)
            out["roofline_hbm_kernels"] = hbm_kernels(model, pool[0], B, torch, ops)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        try:                                     # RCCL leaves a version banner in C stdio: out with it BEFORE the one JSON line
            import ctypes as _ct
            _ct.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
