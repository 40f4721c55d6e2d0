"""World-size-2 run of the fused path on ONE GPU (two processes on cuda:0, gloo collectives): the code the driver
launches at N = 2, 4, 8 with RCCL, checked for numerics here -- naiveSyncBN1d inside the fused VFE kernels
(mmdet3d/ops/norm.py:54-86: equal weight per rank), the gradient all-reduce with the 1/world averaging folded into the
fused AdamW pass, identical parameters on every rank after the step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _frames(rank):
    from geomae_amd import synth
    return [torch.as_tensor(synth.lidar_frame(300 + 10 * rank + i, beams=16, n_az=300 + 40 * rank), device="cuda:0")
            for i in range(2)]


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # the box's hostname may not resolve: pair over loopback
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)          # a stuck rank reports where and leaves (the run takes ~5 s)
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GEOMAE_SIDE_STREAMS", "3")          # two ranks on one GPU: see geomae_amd.ops.side_streams
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        import geomae_amd
        from geomae_amd.configs import mae_sst_model
        from geomae_amd.train import Trainer
        torch.cuda.set_device(0)
        torch.manual_seed(7)                                   # same initial weights on both ranks
        cfg = mae_sst_model(encoder_num_blocks=1, decoder_num_blocks=1)
        cfg["backbone"]["compute_dtype"] = "bf16"
        fused = geomae_amd.build_model(cfg).cuda().train()
        import geomae_oracle as O
        fused.load_state_dict(O.make_params(7, 1, 1), strict=False)       # the weights of tests/golden/g_syncbn_w2.npz
        # sections 1, 3, 4, 5 hold the voxel encoder's cross-rank BatchNorm to the REFERENCE's fixture at fp32 tolerances
        # (running statistics to 1e-5): they run its fp32-grade layer-1 products.  The plain-bf16 products the bf16 step
        # uses (DynamicScatterVFE.compute_dtype, tests/test_gpu_parity.py::test_vfe_plain_bf16_layer1_products) go through
        # the same exchanges; section 2's training steps run them at world size 2.
        assert fused.voxel_encoder.compute_dtype == "bf16"     # (copied from the backbone by the detector)
        fused.voxel_encoder.compute_dtype = "fp32"
        fused_init = copy.deepcopy(fused)                      # the initial state, for the reference run of section 4
        composed = copy.deepcopy(fused)
        composed.voxel_encoder.use_fused = False               # torch ops + the NaiveSyncBatchNorm1d module
        fresh = copy.deepcopy(fused)                           # untouched running statistics, for section 3
        fresh2 = copy.deepcopy(fused)                          # ... and for section 4
        pts = _frames(rank)

        # ---- 1. fused VFE with cross-rank BatchNorm statistics vs the composed module path (fp32 both)
        outs = []
        for m in (fused, composed):
            voxels, coors, _, _ = m.voxelize_all(pts)
            from geomae_amd import ops
            seg = ops.pillar_segment(coors, len(pts), m.grid_size)
            vf, _ = m.voxel_encoder(voxels, coors, seg=seg)
            w = torch.randn(vf.shape, generator=torch.Generator().manual_seed(rank)).cuda()
            (vf * w).sum().backward()
            outs.append((vf.detach().clone(), {k: p.grad.clone() for k, p in m.voxel_encoder.named_parameters()},
                         {k: b.clone() for k, b in m.voxel_encoder.named_buffers() if "running" in k}))
            for p in m.parameters():
                p.grad = None
        (vf_f, g_f, rb_f), (vf_c, g_c, rb_c) = outs
        assert torch.allclose(vf_f, vf_c, rtol=1e-4, atol=2e-4), float((vf_f - vf_c).abs().max())
        for k in g_f:
            rel = float((g_f[k] - g_c[k]).norm() / g_c[k].norm().clamp(min=1e-12))
            assert rel < 1e-2, (k, rel)          # dW1 is contracted from bf16 copies of dy1 and g (dw_kernel)
        for k in rb_f:
            assert torch.allclose(rb_f[k], rb_c[k], rtol=1e-5, atol=1e-6), k
        # ---- 1b. both against the REFERENCE: its DynamicScatterVFE with the real NaiveSyncBatchNorm1d, run at world
        #          size 2 over gloo on these same frames (tests/golden/g_syncbn_w2.npz part 2, oracle/make_golden_syncbn.py)
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_syncbn_w2.npz"))
        G = lambda k: torch.as_tensor(gold[f"r{rank}.{k}"]).cuda()
        assert int(gold[f"r{rank}.v_n_points"]) == sum(p.shape[0] for p in pts)
        worst = {}
        for tag, (vf_x, g_x, rb_x) in (("fused", outs[0]), ("composed", outs[1])):
            assert vf_x.shape[0] == gold[f"r{rank}.v_coors"].shape[0]
            assert torch.allclose(vf_x[::4], G("v_rows"), rtol=1e-4, atol=2e-4), (tag, float((vf_x[::4] - G("v_rows")).abs().max()))
            assert torch.allclose(vf_x.double().sum(0), G("v_colsum"), rtol=1e-4, atol=1e-2), tag
            for k in g_x:
                rel = float((g_x[k] - G("v_grad." + k)).norm() / G("v_grad." + k).norm().clamp(min=1e-12))
                worst[(tag, k)] = rel
                # composed (fp32 torch ops): 2e-3.  Fused: dW1 is contracted from bf16 copies of dy1 and g (dw_kernel),
                # and the layer-0 gradients pass through the bf16x3 split GEMMs of layer 1 (measured 1.5e-3 ... 3.0e-3 run to run: arg-max ties)
                assert rel < ((1e-2 if k.endswith("1.linear.weight") else 8e-3) if tag == "fused" else 2e-3), (tag, k, rel)
            for k in rb_x:
                assert torch.allclose(rb_x[k], G("v_buf." + k), rtol=1e-5, atol=1e-6), (tag, k)
        for m_ in (fused, composed):          # the cross-rank branch leaves the batch counter alone (ops/norm.py:58-86)
            assert all(int(b) == 0 for k, b in m_.voxel_encoder.named_buffers() if k.endswith("num_batches_tracked"))
        if os.environ.get("GEOMAE_TEST_VERBOSE"):
            print(f"rank {rank}: VFE vs reference world-2 fixture, worst gradient differences:",
                  {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:4]}, flush=True)

        # ---- 2. two full training steps: every rank ends with the same parameters, and they are what the composed
        #         optimizer path (all-reduce + mul 1/world + clip + AdamW) produces from the same gradients
        # (first: the gradient buffer the optimizer reads.  The explicit schedule starts the all-reduce of the early
        #  segment from the geometry stream while the encoder backward runs; the autograd path reduces everything after
        #  its backward on one stream.  Same sums up to the atomics / bf16 noise floor of tests/test_gpu_parity.py.)
        fused.voxel_encoder.compute_dtype = "bf16"            # (the step's own mode: plain bf16 layer-1 products)
        auto = copy.deepcopy(fused)
        tr, tr_auto = Trainer(fused), Trainer(auto)
        tr_auto.explicit_schedule = False
        taps = {}
        tr.on_reduced_grad = lambda g: taps.setdefault("explicit", g.clone())
        tr_auto.on_reduced_grad = lambda g: taps.setdefault("autograd", g.clone())
        tr_auto.train_step(pts)
        losses, gnorm = tr.train_step(pts)
        tr.on_reduced_grad = None
        assert len(tr.flat.segments) == 3                      # i.e. both early all-reduces did run
        assert tr.engine is not None and tr.engine.world == 2  # the C step engine, SyncBN + early exchanges through its hook
        worst = (0.0, "")
        for name, off, p in zip(tr.flat.names, tr.flat.offsets, tr.flat.params):
            ge, ga = taps["explicit"][off:off + p.numel()], taps["autograd"][off:off + p.numel()]
            d = float((ge - ga).norm() / ga.norm().clamp(min=1e-12))
            worst = max(worst, (d, name))
            assert d < 1e-2, (name, d)                         # measured: <= 1.5e-3 (VFE layer 0, fp32 atomics order)
        if os.environ.get("GEOMAE_TEST_VERBOSE"):
            print(f"rank {rank}: largest explicit-vs-autograd gradient difference {worst[0]:.2e} ({worst[1]})", flush=True)
        for _ in range(2):
            losses, gnorm = tr.train_step(pts)
        assert all(torch.isfinite(v) for v in losses.values()) and torch.isfinite(gnorm)
        mine = tr.flat.flat.clone()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert torch.equal(gathered[0], gathered[1])
        gn = torch.stack([gnorm.detach().float().reshape(())]).cuda()
        both = [torch.zeros_like(gn) for _ in range(world)]
        dist.all_gather(both, gn)
        assert torch.equal(both[0], both[1])                   # the clip coefficient is computed from the reduced gradient

        # the composed optimizer path (all-reduce, * 1/world, clip, AdamW in torch ops) from the SAME local gradients
        # (a second backward would differ in the last bits -- float atomics -- and Adam's first steps are sign-like)
        from geomae_amd.train import FlatAdamW, FlatParams, allreduce_gradients, clip_grad_norm
        def toy():
            torch.manual_seed(11)
            return torch.nn.Sequential(torch.nn.Linear(96, 200), torch.nn.LayerNorm(200), torch.nn.Linear(200, 33)).cuda()
        fa, fb = FlatParams(toy(), no_decay_keys=("1.",)), FlatParams(toy(), no_decay_keys=("1.",))
        oa, ob = FlatAdamW(fa, lr=1e-3), FlatAdamW(fb, lr=1e-3)
        for step in range(3):
            g = torch.randn(fa.grad.shape, generator=torch.Generator().manual_seed(50 + 10 * step + rank)).cuda() * (1 + step)
            fa.grad.copy_(g)
            fb.grad.copy_(g)
            dist.all_reduce(fa.grad, op=dist.ReduceOp.SUM)
            na = oa.fused_clip_step(10.0, 1.0 / world, zero_grad=True)
            allreduce_gradients(fb)
            nb = clip_grad_norm(fb, 10.0)
            ob.step()
            assert abs(float(na) - float(nb)) <= 2e-6 * float(nb)
            assert torch.allclose(fa.flat, fb.flat, rtol=2e-6, atol=1e-8), float((fa.flat - fb.flat).abs().max())
        # ---- 3. the C step ENGINE's cross-rank BatchNorm against the REFERENCE: one engine step at world size 2 (ranks
        #         with different point counts; the first layer's statistics travel a step ahead as rank-averaged feature
        #         moments, the second layer's in line) must leave the running statistics the reference's
        #         NaiveSyncBatchNorm1d left on these frames and weights (the VFE forward does not depend on the mask)
        tr_fresh = Trainer(fresh)
        tr_fresh.train_step(pts)
        torch.cuda.synchronize()
        assert tr_fresh.engine is not None and tr_fresh.engine.world == 2 and "featmom" in tr_fresh.engine.sync
        for k, b in fresh.voxel_encoder.named_buffers():
            if "running" in k:
                assert torch.allclose(b, G("v_buf." + k), rtol=1e-5, atol=1e-6), (k, float((b - G("v_buf." + k)).abs().max()))
            if k.endswith("num_batches_tracked"):
                assert int(b) == 0, k
        # ---- 4. workspace growth on ONE rank only (rank 1's engine is created for 200 pillars: its first step returns
        #         GEOMAE_ERR_WORKSPACE, the wrapper re-creates the engine and re-submits the batch).  The batch's feature
        #         moments were exchanged at its first submission; the re-submission must not raise a second all-reduce
        #         that rank 0 never issues (every later SyncBN collective would pair with the wrong one, or hang).  Same
        #         frames and weights as section 3: the running statistics must come out the same, on both ranks.
        tr_grow = Trainer(fresh2)
        eng = tr_grow.get_engine()
        assert eng is not None and eng.world == 2
        if rank == 1:
            eng.max_pillars = 200
        for _ in range(2):
            losses_g, gnorm_g = tr_grow.train_step(pts)
        torch.cuda.synchronize()
        if rank == 1:
            assert eng.max_pillars > 200                        # i.e. the growth path did run
        assert all(torch.isfinite(v) for v in losses_g.values()) and torch.isfinite(gnorm_g)
        mine = tr_grow.flat.flat.clone()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert torch.equal(gathered[0], gathered[1])
        tr_ref = Trainer(copy.deepcopy(fused_init))
        for _ in range(2):
            tr_ref.train_step(pts)
        torch.cuda.synchronize()
        for (k, b), (_, b2) in zip(fresh2.voxel_encoder.named_buffers(), tr_ref.model.voxel_encoder.named_buffers()):
            if "running" in k:
                assert torch.allclose(b, b2, rtol=1e-5, atol=1e-6), (k, float((b - b2).abs().max()))
        # ---- 5. the same growth on a PIPELINED batch: the big batch is handed over as step 1's next_points, so its
        #         feature-moment all-reduce was issued on the decoder-B stream during step 1; step 2 then overflows on rank 1
        #         only.  The moments carried into the re-submission must be the REDUCED ones (the wrapper synchronises
        #         before cloning them): running statistics as a run that never overflowed, parameters equal on both ranks.
        from geomae_amd import synth
        small = [torch.as_tensor(synth.lidar_frame(700 + 10 * rank + i, beams=8, n_az=200), device="cuda:0") for i in range(2)]
        big = pts

        def pillars(frames):
            v_, c_, _, _ = fused_init.voxelize_all(frames)
            return int(ops.pillar_segment(c_, len(frames), fused_init.grid_size).V)
        v_small, v_big = pillars(small), pillars(big)
        assert v_big > v_small + 64, (v_small, v_big)
        tr_pipe = Trainer(copy.deepcopy(fused_init))
        eng = tr_pipe.get_engine()
        cap = (v_small + v_big) // 2
        if rank == 1:
            eng.max_pillars = cap
        tr_pipe.train_step(small, next_points=big)
        if rank == 1:
            assert eng.max_pillars == cap                       # step 1 fitted; the big batch is pending
        losses_p, gnorm_p = tr_pipe.train_step(big)
        torch.cuda.synchronize()
        if rank == 1:
            assert eng.max_pillars > cap                        # step 2 overflowed and grew
        assert all(torch.isfinite(v) for v in losses_p.values()) and torch.isfinite(gnorm_p)
        mine = tr_pipe.flat.flat.clone()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert torch.equal(gathered[0], gathered[1])
        tr_ref2 = Trainer(copy.deepcopy(fused_init))
        tr_ref2.train_step(small, next_points=big)
        tr_ref2.train_step(big)
        torch.cuda.synchronize()
        for (k, b), (_, b2) in zip(tr_pipe.model.voxel_encoder.named_buffers(), tr_ref2.model.voxel_encoder.named_buffers()):
            if "running" in k:
                assert torch.allclose(b, b2, rtol=1e-5, atol=1e-6), (k, float((b - b2).abs().max()))
        torch.save(dict(ok=True), os.path.join(tmp, f"ok{rank}.pt"))
        faulthandler.cancel_dump_traceback_later()
    except BaseException:
        # a rank that fails while its peer sits in a collective would leave both hanging (the peer in the collective,
        # this one in destroy_process_group): report and leave at once, mp.spawn then stops the peer
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    dist.destroy_process_group()


def test_world_size_2_fused_path_on_one_gpu(tmp_path):
    assert torch.cuda.is_available()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}.pt") for r in range(2))
