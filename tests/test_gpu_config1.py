"""BASELINE configs[0] as a full HIP step: 0.5 m pillars -> fp32 grid 205, which is NOT a multiple
of the 12-pillar window (18 window slots per axis, the last one cut), sub-voxels 0.25 / 0.125 m, SST-tiny 1 + 1 blocks.
Both the autograd path (forward_train) and the C step engine against the fixture the reference produced at that
geometry (tests/golden/g_pipeline_c1.npz): six losses, every gradient norm, five full gradient tensors."""
import os
import sys

import numpy as np
import pytest
import torch

import geomae_oracle as O
from fullsize_cases import C1, CASES_C1, make_frames

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
FULL = (("grad_vfe0", "voxel_encoder.vfe_layers.0.linear.weight"), ("grad_mask_token", "backbone.mask_token"),
        ("grad_pred_top_w", "backbone.decoder_pred_top.weight"),
        ("grad_enc5_ffn_b", "backbone.encoder_blocks.0.encoder_list.1.linear1.bias"),
        ("grad_dec_out_w", "backbone.decoder_centroid_blocks.0.encoder_list.1.win_attn.self_attn.out_proj.weight"))
# (loss, gradient norm, full-gradient Frobenius); the fp32 composed path / the bf16 fused path and engine
TOL = {"fp32": (1e-3, 1e-3, 1.5e-3), "bf16": (1e-2, 3.5e-2, 5.5e-2)}      # fp32 measured (LiDAR cloud): 4.4e-4 / 2.9e-4 / 4.9e-4
# The 16 k UNIFORM cloud at 0.5 m leaves 17.8 % of the masked pillars (c1u; 16.4 % in the batch c1b, 3.4 % in the LiDAR
# cloud) with <= 2 occupied med cells in their 3 x 3 neighbourhood: the scatter matrix has rank <= 1, its two smallest
# eigenvalues coincide, and the "normal" is whatever vector of that null space the solver returns (LAPACK gesdd in the
# reference, ssl.py:598-602; a Jacobi sweep here).  The normal target, hence loss_curv_around and the gradients of the
# density decoder that regresses it, are defined only up to that choice on those rows: bounds for them on the fp32
# path are set from the measured 5.6e-3 (loss) / 6.9e-3 (gradient norm) / 3.6e-3 (the five other losses and the centroid
# decoder keep the tight ones).
ILL = {"c1u": (1.5e-2, 1.5e-2, 8e-3), "c1b": (1.5e-2, 1.5e-2, 8e-3)}


def _model(compute_dtype):
    import geomae_amd
    from geomae_amd.configs import mae_sst_model
    enc, dec = C1["blocks"]
    cfg = mae_sst_model(encoder_num_blocks=enc, decoder_num_blocks=dec, voxel_size=C1["top"], sub_voxel_size_low=C1["low"],
                        sub_voxel_size_med=C1["med"], point_cloud_range=C1["range"], grid_size=C1["grid"])
    cfg["backbone"]["output_shape"] = list(C1["grid"][1:])
    cfg["backbone"]["compute_dtype"] = compute_dtype
    model = geomae_amd.build_model(cfg).cuda()
    missing = model.load_state_dict(O.make_params(7, enc, dec), strict=False)
    assert not missing.unexpected_keys
    return model.train()


def _check(tag, model, losses, K, tols, ill=None):
    tol_l, tol_n, tol_f = tols
    ref = dict(zip([str(n) for n in K("loss_names")], [float(v) for v in K("loss_vals")]))
    assert set(losses) == set(ref)
    errs = {k: abs(v - ref[k]) / max(1.0, abs(ref[k])) for k, v in losses.items()}
    if ill is not None:           # the normal regression on rank-deficient neighbourhoods: its own bound (see ILL)
        assert errs.pop("loss_curv_around") <= ill[0], (errs, losses, ref)
        tol_n, tol_f = max(tol_n, ill[1]), max(tol_f, ill[2])
    e_loss = max(errs.values())
    named = dict(model.named_parameters())
    gn = dict(zip([str(n) for n in K("grad_names")], [float(v) for v in K("grad_norms")]))
    assert set(gn) == set(named)
    e_norm, worst = 0.0, ""
    for k, p in named.items():
        e = abs(float(p.grad.double().norm()) - gn[k]) / max(gn[k], 1e-2)
        if ill is not None and "centroid" in k:      # the centroid decoder never sees the normal targets
            assert e <= tols[1], (k, e)
        if e > e_norm:
            e_norm, worst = e, k
    e_full = {}
    for key, name in FULL:
        a, b = named[name].grad.detach().double().cpu().numpy(), K(key).astype(np.float64)
        e_full[key] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"\nconfig 1 {tag}: loss err {e_loss:.2e}, worst grad-norm err {e_norm:.2e} ({worst}), Frobenius "
          f"{({k: f'{v:.1e}' for k, v in e_full.items()})}", flush=True)
    assert e_loss <= tol_l, (e_loss, losses, ref)
    assert e_norm <= tol_n, (e_norm, worst)
    assert max(e_full.values()) <= tol_f, e_full


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", list(CASES_C1))
def test_forward_train_at_config1_geometry(golden_dir, case, compute_dtype):
    g = np.load(os.path.join(golden_dir, "g_pipeline_c1.npz"))
    K = lambda k: g[f"{case}.{k}"]
    frames = make_frames(CASES_C1[case][1])
    model = _model(compute_dtype)
    pts = [torch.as_tensor(f, device="cuda") for f in frames]
    # voxel coordinates and their order: bit-exact against the reference's own voxelizer + torch.unique
    from geomae_amd import ops
    voxels, coors, _, _ = model.voxelize_all(pts)
    seg = ops.pillar_segment(coors, len(pts), model.grid_size)
    assert np.array_equal(seg.voxel_coors[:seg.V].cpu().numpy(), K("voxel_coors").astype(np.int32))
    ik = torch.as_tensor(K("ids_keep").astype(np.int64), device="cuda")
    im = torch.as_tensor(K("ids_mask").astype(np.int64), device="cuda")
    losses = model.forward_train(pts, None, ids_keep=ik, ids_mask=im)
    sum(losses.values()).backward()
    _check(f"{case} forward_train {compute_dtype}", model, {k: float(v.detach()) for k, v in losses.items()}, K, TOL[compute_dtype],
           ILL.get(case))


@pytest.mark.parametrize("case", list(CASES_C1))
def test_engine_step_at_config1_geometry(golden_dir, case):
    from geomae_amd.engine import PretrainEngine
    from geomae_amd.train import Trainer
    g = np.load(os.path.join(golden_dir, "g_pipeline_c1.npz"))
    K = lambda k: g[f"{case}.{k}"]
    frames = make_frames(CASES_C1[case][1])
    model = _model("bf16")
    tr = Trainer(model)
    eng = PretrainEngine(model, tr.flat, tr.opt, 10.0)
    pts = [torch.as_tensor(f, device="cuda") for f in frames]
    tr.flat.zero_grad()
    losses, _ = eng.step(pts, pts, 1e-5, run_optimizer=False, ids_keep=torch.as_tensor(K("ids_keep").astype(np.int64)),
                         ids_mask=torch.as_tensor(K("ids_mask").astype(np.int64)))
    torch.cuda.synchronize()
    assert eng.last_sizes()["V"] == int(K("V"))
    _check(f"{case} engine bf16", model, {k: float(losses[i]) for i, k in enumerate(model.LOSS_KEYS)}, K, TOL["bf16"], ILL.get(case))
    # and two real training steps (random mask, optimizer) stay finite at this geometry
    for _ in range(2):
        l2, gnorm = tr.train_step(pts, next_points=pts)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in l2.values()) and torch.isfinite(gnorm)
