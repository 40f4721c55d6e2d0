"""The world > 1 schedule of the step engine executed on RCCL ("nccl" backend) with ONE rank on one GPU.

Everything the 8-GPU run does differently from N = 1 -- the engine's hooks (naiveSyncBN1d's four [2C] all-reduces on the
SyncBN group, mmdet3d/ops/norm.py:9-24,54-86; the two early gradient-segment all-reduces started from the geometry
stream), the late segment, `w.wait()`, the optimizer as a second C call -- runs here against a real RCCL communicator,
where collectives are STREAM-ordered (gloo's are host-blocking round trips: tests/test_gpu_multirank.py cannot see an
ordering bug).  With one rank every all-reduce is the identity, so the forced path must reproduce the plain N = 1
engine: losses, gradient norm, BatchNorm running statistics, parameters after three optimizer steps.
(tools/dist_train.sh:8-9 launches the reference the same way: one process per GPU, NCCL.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    try:
        import copy
        import geomae_amd
        from geomae_amd import engine as E, synth
        from geomae_amd.configs import mae_sst_model
        from geomae_amd.train import Trainer, exchange_mode
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        torch.manual_seed(5)
        cfg = mae_sst_model(encoder_num_blocks=2, decoder_num_blocks=1)
        cfg["backbone"]["compute_dtype"] = "bf16"
        plain = geomae_amd.build_model(cfg).cuda().train()
        forced = copy.deepcopy(plain)
        pool = [[torch.as_tensor(synth.lidar_frame(900 + 10 * i + b, beams=16, n_az=450 + 30 * b), device=dev)
                 for b in range(2)] for i in range(3)]

        os.environ.pop("GEOMAE_FORCE_EXCHANGE", None)
        assert exchange_mode() == (1, False)
        tr_p = Trainer(plain)
        os.environ["GEOMAE_FORCE_EXCHANGE"] = "1"
        assert exchange_mode() == (1, True)
        tr_f = Trainer(forced)                                  # builds the SyncBN group, primes the RCCL streams
        from geomae_amd import ops
        assert ops.BN_GROUP is not None
        calls = []
        eng_f = tr_f.get_engine()
        assert eng_f.exchange and eng_f.world == 1
        body = eng_f._hook_body

        def counting(what, stream):
            calls.append((what, int(stream or 0)))
            return body(what, stream)
        eng_f._hook_body = counting
        taps = []
        tr_f.on_reduced_grad = lambda g: taps.append(float(g.double().norm()))

        hist = []
        for i in range(3):
            os.environ.pop("GEOMAE_FORCE_EXCHANGE", None)
            lp, gp = tr_p.train_step(pool[i], next_points=pool[(i + 1) % 3])
            lp = torch.stack([lp[k] for k in plain.LOSS_KEYS]).clone()
            os.environ["GEOMAE_FORCE_EXCHANGE"] = "1"
            lf, gf = tr_f.train_step(pool[i], next_points=pool[(i + 1) % 3])
            lf = torch.stack([lf[k] for k in plain.LOSS_KEYS]).clone()
            torch.cuda.synchronize()
            hist.append((lp, lf, float(gp), float(gf)))
            if i == 0:      # same parameters on both sides during step 0: the running statistics moved identically
                for (k, a), (_, b) in zip(forced.named_buffers(), plain.named_buffers()):
                    if "num_batches" in k:        # (the cross-rank branch leaves the counter alone, ops/norm.py:58-86)
                        continue
                    # ... and feeds the BIASED batch variance into running_var (ops/norm.py:64-76) where nn.BatchNorm1d
                    # uses the unbiased one: a factor N / (N - 1) = 1 + 7e-5 on the increment at N = 14 k points
                    assert torch.allclose(a.float(), b.float(), rtol=3e-4 if "running_var" in k else 1e-5, atol=1e-6), k
        assert tr_p.get_engine().exchange is False
        # per step: 3 SyncBN exchanges in line + 2 early segment exchanges + the NEXT batch's feature moments (layer-0
        # statistics, exchanged a step ahead) behind them; the first batch's moments inside its submission.  Each hook
        # is handed the stream it must be ordered on.
        assert len(calls) == 1 + 3 * 6, calls
        assert calls[0][0] == E.HOOK_FEAT_MOMENTS0
        for i in range(3):
            per_step = [w for w, _ in calls[1 + 6 * i:7 + 6 * i]]
            assert per_step == [E.HOOK_BN_FWD1, E.HOOK_GRADS_EARLY, E.HOOK_GRADS_ENCODER, E.HOOK_BN_BWD1, E.HOOK_BN_BWD0,
                                E.HOOK_FEAT_MOMENTS1 if i % 2 == 0 else E.HOOK_FEAT_MOMENTS0], (i, per_step)
        main = torch.cuda.current_stream(dev).cuda_stream
        for j, (w, s) in enumerate(calls):
            want = eng_f.geo.cuda_stream if w in (E.HOOK_GRADS_EARLY, E.HOOK_GRADS_ENCODER) else main
            if w in (E.HOOK_FEAT_MOMENTS0, E.HOOK_FEAT_MOMENTS1) and j > 0:
                want = eng_f.aux.cuda_stream               # (stage 1 of the next batch runs on the decoder-B stream)
            assert s == want, (w, s, want)
        for i, (lp, lf, gp, gf) in enumerate(hist):
            # the run-to-run noise of either path: 2e-4 on the losses (tools/archive/engine_noise.py); bound = 6x
            assert torch.allclose(lf, lp, rtol=1.5e-3, atol=1e-6), (i, lf, lp)
            assert abs(gf - gp) <= 2e-3 * gp, (i, gf, gp)
            assert abs(taps[i] - gf) <= 1e-4 * gf, (i, taps[i], gf)      # the optimizer read the exchanged buffer
        d = float((tr_f.flat.flat - tr_p.flat.flat).abs().max())
        assert d <= 3 * 2 * 1e-5, d                              # AdamW's first steps are sign-like: |update| ~ lr
        for (k, a), (_, b) in zip(forced.named_buffers(), plain.named_buffers()):
            if "num_batches" in k:
                continue
            # (after three sign-like AdamW steps the two sides' weights differ by ~lr: the statistics follow)
            assert torch.allclose(a.float(), b.float(), rtol=2e-3, atol=1e-5), k
        assert tr_f.opt.step_count == 3 and tr_f.get_engine().last_sizes()["optimizer_steps"] == 3
        torch.save(dict(ok=True, losses=[h[1].cpu() for h in hist]), os.path.join(tmp, "ok.pt"))
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)


def test_forced_exchange_on_rccl_matches_plain_engine(tmp_path):
    assert torch.cuda.is_available()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok.pt")
