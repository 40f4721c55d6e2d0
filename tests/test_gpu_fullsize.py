"""BASELINE.json configs 2, 3 and 4 at FULL size through the HIP path against summary fixtures from the reference
(tests/golden/g_fullsize.npz, oracle/make_golden_fullsize.py): 6 losses, every parameter's gradient norm, five full
gradient tensors (relative Frobenius).  Frames are regenerated from seeds; the reference's mask indices are injected.
Tolerances are ~2x the measured maxima printed by this test (GEOMAE_TEST_VERBOSE=1)."""
import os
import sys

import numpy as np
import pytest
import torch

import geomae_oracle as O

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))

GEOM_WAYMO = dict(voxel_size=(0.32, 0.32, 6), sub_voxel_size_low=(0.08, 0.08, 0.75), sub_voxel_size_med=(0.16, 0.16, 1.5),
                  point_cloud_range=(-74.88, -74.88, -2.0, 74.88, 74.88, 4.0), grid_size=(1, 468, 468))
FULL = ("grad_vfe0", "voxel_encoder.vfe_layers.0.linear.weight"), ("grad_mask_token", "backbone.mask_token"), \
    ("grad_pred_top_w", "backbone.decoder_pred_top.weight"), \
    ("grad_enc5_ffn_b", "backbone.encoder_blocks.5.encoder_list.1.linear1.bias"), \
    ("grad_dec_out_w", "backbone.decoder_centroid_blocks.1.encoder_list.1.win_attn.self_attn.out_proj.weight")
# (loss tolerance, gradient-norm tolerance, full-gradient relative Frobenius tolerance) per compute dtype
# measured maxima over c2 / c3 / c4 (GEOMAE_TEST_VERBOSE=1): fp32 composed path 5.2e-5 / 1.9e-4 / 3.5e-4 (round 3: fp32
# attention core in it; with the bf16 attention kernel 5.0e-4 / 6.8e-4 / 2.4e-3); bf16 fused path 4.2e-3 / 1.7e-2 / 2.7e-2
# (the VFE layer-0 weight at 10 sweeps: 260 k points reduced through bf16x3 products)
TOL = {"fp32": (1.5e-4, 5e-4, 1e-3), "bf16": (1e-2, 3.5e-2, 5.5e-2)}


def _frames(case):
    from fullsize_cases import CASES                            # the seeds / generator arguments only
    from geomae_amd import synth
    return [synth.lidar_frame(**kw) for kw in CASES[case][1]]


def _model(case, compute_dtype):
    import geomae_amd
    from geomae_amd.configs import mae_sst_model
    cfg = mae_sst_model(**(GEOM_WAYMO if case == "c4" else {}))
    if case == "c4":
        cfg["backbone"]["output_shape"] = [468, 468]
    cfg["backbone"]["compute_dtype"] = compute_dtype
    model = geomae_amd.build_model(cfg).cuda()
    missing = model.load_state_dict(O.make_params(7, 6, 2), strict=False)
    assert not missing.unexpected_keys
    return model.train()


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["c2", "c3", "c4"])
def test_full_size_step_matches_reference_summary(golden_dir, case, compute_dtype):
    g = np.load(os.path.join(golden_dir, "g_fullsize.npz"))
    K = lambda k: g[f"{case}.{k}"]
    frames = _frames(case)
    assert [f.shape[0] for f in frames] == list(K("n_points"))
    model = _model(case, compute_dtype)
    pts = [torch.as_tensor(f, device="cuda") for f in frames]
    ik = torch.as_tensor(K("ids_keep").astype(np.int64), device="cuda")
    im = torch.as_tensor(K("ids_mask").astype(np.int64), device="cuda")
    losses = model.forward_train(pts, None, ids_keep=ik, ids_mask=im)
    ref = dict(zip([str(n) for n in K("loss_names")], K("loss_vals")))
    assert set(losses) == set(ref)
    tol_l, tol_n, tol_f = TOL[compute_dtype]
    e_loss = max(abs(float(v.detach()) - ref[k]) / max(1.0, abs(ref[k])) for k, v in losses.items())
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    gn = dict(zip([str(n) for n in K("grad_names")], K("grad_norms")))
    assert set(gn) == set(named)
    e_norm, worst = 0.0, ""
    for k, p in named.items():
        e = abs(float(p.grad.double().norm()) - gn[k]) / max(gn[k], 1e-2)
        if e > e_norm:
            e_norm, worst = e, k
    e_full = {}
    for key, name in FULL:
        a, b = named[name].grad.detach().double().cpu().numpy(), K(key).astype(np.float64)
        e_full[key] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    if os.environ.get("GEOMAE_TEST_VERBOSE"):
        print(f"\n{case} {compute_dtype}: V={int(K('V'))} loss err {e_loss:.2e}, worst grad-norm err {e_norm:.2e} ({worst}), "
              f"full-gradient Frobenius errs {({k: f'{v:.1e}' for k, v in e_full.items()})}", flush=True)
    assert e_loss <= tol_l, (e_loss, {k: (float(v), ref[k]) for k, v in losses.items()})
    assert e_norm <= tol_n, (e_norm, worst)
    assert max(e_full.values()) <= tol_f, e_full
    assert all(torch.isfinite(p.grad).all() for p in named.values())


def test_full_size_engine_step_runs_on_every_workload():
    """The C step engine on configs 3 and 4 (10-sweep frame, Waymo geometry): finite losses, sizes as the fixture's."""
    from geomae_amd.train import Trainer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_fullsize.npz"))
    for case in ("c3", "c4"):
        model = _model(case, "bf16")
        tr = Trainer(model)
        pts = [torch.as_tensor(f, device="cuda") for f in _frames(case)]
        losses, gnorm = tr.train_step(pts, next_points=pts)
        losses2, _ = tr.train_step(pts)
        torch.cuda.synchronize()
        s = tr.engine.last_sizes()
        assert s["V"] == int(g[f"{case}.V"]) and s["N"] == int(g[f"{case}.n_points"].sum())
        assert all(torch.isfinite(v) for v in losses2.values()) and torch.isfinite(gnorm)
        # a random 70 % mask of the same frame: the losses are close to the fixture's (another mask of the same data)
        ref = dict(zip([str(n) for n in g[f"{case}.loss_names"]], g[f"{case}.loss_vals"]))
        for k, v in losses.items():
            assert abs(float(v) - ref[k]) <= 0.15 * max(1.0, abs(ref[k])), (case, k, float(v), ref[k])


def test_engine_at_config3_per_gpu_load():
    """BASELINE configs[2] at its REAL per-GPU load: B = 4 ten-sweep frames (~1.0 M points, ~84 k pillars) through the C
    step engine -- the batch the 8-GPU metric is quoted on (`bench.py --workload nuscenes10`).  The pillar coordinates the
    engine's stage 1 leaves are bit-equal to the oracle's unique rows (= torch.unique(dim=0), sst_ops.py:15) of the
    oracle's voxelization; sizes; the mask is a partition with int(L * 0.3) kept pillars per sample; finite losses and
    gradient norm over two optimizer steps (the second on the batch handed over as next_points)."""
    import geomae_oracle as O
    from geomae_amd import synth
    from geomae_amd.train import Trainer
    frames = [synth.lidar_frame(3100 + i, sweeps=10) for i in range(4)]
    from fullsize_cases import NUS
    _, coors = O.voxelize_batch(frames, NUS["top"], NUS["range"])
    want_vc = O.unique_rows(coors)[0]
    model = _model("c3", "bf16")
    tr = Trainer(model)
    pts = [torch.as_tensor(f, device="cuda") for f in frames]
    losses, gnorm = tr.train_step(pts, next_points=pts)
    torch.cuda.synchronize()
    eng = tr.engine
    s = eng.last_sizes()
    assert s["N"] == sum(f.shape[0] for f in frames) and s["N"] > 900_000
    assert s["V"] == want_vc.shape[0]
    got_vc = eng.last_voxel_coors().cpu().numpy()
    assert np.array_equal(got_vc, want_vc)                       # bit-exact pillar set AND order
    ik, im = (t.cpu().numpy() for t in eng.last_ids())
    per_sample = np.bincount(want_vc[:, 0], minlength=4)
    assert ik.size == sum(int(L * (1 - model.random_mask_ratio)) for L in per_sample) and ik.size + im.size == s["V"]
    assert np.array_equal(np.sort(np.concatenate([ik, im])), np.arange(s["V"]))
    assert all(torch.isfinite(v) for v in losses.values()) and torch.isfinite(gnorm)
    losses2, gnorm2 = tr.train_step(pts)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in losses2.values()) and torch.isfinite(gnorm2)
    assert tr.opt.step_count == 2
