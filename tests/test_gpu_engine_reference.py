"""The BENCHMARKED orchestration -- geomae_pretrain_step (csrc/engine.hip), with everything only the engine does:
dead-row skipping in the decoders' last layer (set_first_live_row), the split heads kernel, cross-step stage 1,
workspace carving, three streams -- pinned DIRECTLY to fixtures produced by the reference
(MultiSubVoxelDynamicVoxelNetSSL.forward_train + backward, ssl.py:126-242; oracle/make_golden*.py).

The reference's own mask (its torch.randperm draw, unsorted order) is injected with geomae_pretrain_set_mask; the
step runs without its optimizer pass so that the gradient buffer can be read.  Compared: the six losses, the gradient
norm of EVERY parameter, full gradient tensors (relative Frobenius), at the same bf16 bounds the autograd path is held
to (tests/test_gpu_parity.py::test_forward_train_losses_and_grads, tests/test_gpu_fullsize.py)."""
import os
import sys

import numpy as np
import pytest
import torch

import geomae_oracle as O
from geomae_amd import synth

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))

GEOM_WAYMO = dict(voxel_size=(0.32, 0.32, 6), sub_voxel_size_low=(0.08, 0.08, 0.75), sub_voxel_size_med=(0.16, 0.16, 1.5),
                  point_cloud_range=(-74.88, -74.88, -2.0, 74.88, 74.88, 4.0), grid_size=(1, 468, 468))
SMALL_FULL = (("grad_pred_top_w", "backbone.decoder_pred_top.weight"), ("grad_vfe0", "voxel_encoder.vfe_layers.0.linear.weight"),
              ("grad_mask_token", "backbone.mask_token"),
              ("grad_enc0_inproj_bias", "backbone.encoder_blocks.0.encoder_list.0.win_attn.self_attn.in_proj_bias"))
BIG_FULL = (("grad_vfe0", "voxel_encoder.vfe_layers.0.linear.weight"), ("grad_mask_token", "backbone.mask_token"),
            ("grad_pred_top_w", "backbone.decoder_pred_top.weight"),
            ("grad_enc5_ffn_b", "backbone.encoder_blocks.5.encoder_list.1.linear1.bias"),
            ("grad_dec_out_w", "backbone.decoder_centroid_blocks.1.encoder_list.1.win_attn.self_attn.out_proj.weight"))
# (loss, gradient-norm, full-gradient Frobenius): the bounds of the autograd path's bf16 tests
TOL_SMALL = (6e-3, 1.2e-2, 3e-2)
TOL_BIG = (1e-2, 3.5e-2, 5.5e-2)


def _model(geom=None, enc=6, dec=2):
    import geomae_amd
    from geomae_amd.configs import mae_sst_model
    cfg = mae_sst_model(encoder_num_blocks=enc, decoder_num_blocks=dec, **(geom or {}))
    if geom:
        cfg["backbone"]["output_shape"] = list(geom["grid_size"][1:])
    cfg["backbone"]["compute_dtype"] = "bf16"
    model = geomae_amd.build_model(cfg).cuda()
    missing = model.load_state_dict(O.make_params(7, enc, dec), strict=False)
    assert not missing.unexpected_keys
    return model.train()


def _engine_step(model, frames, ids_keep, ids_mask, next_frames=None):
    """One engine step (no optimizer pass) on `frames` under the injected mask -> (losses dict, trainer, engine)."""
    from geomae_amd.engine import PretrainEngine
    from geomae_amd.train import Trainer
    tr = Trainer(model)
    eng = PretrainEngine(model, tr.flat, tr.opt, 10.0)
    pts = [torch.as_tensor(f, device="cuda") for f in frames]
    nxt = [torch.as_tensor(f, device="cuda") for f in next_frames] if next_frames is not None else None
    tr.flat.zero_grad()
    losses, _ = eng.step(pts, nxt, 1e-5, run_optimizer=False, ids_keep=torch.as_tensor(ids_keep.astype(np.int64)),
                         ids_mask=torch.as_tensor(ids_mask.astype(np.int64)))
    torch.cuda.synchronize()
    ik, im = eng.last_ids()
    assert np.array_equal(ik.cpu().numpy(), ids_keep.astype(np.int32)) and np.array_equal(im.cpu().numpy(), ids_mask.astype(np.int32))
    return {k: float(losses[i]) for i, k in enumerate(model.LOSS_KEYS)}, tr, eng


def _compare(tag, model, losses, ref_losses, ref_norms, full, tols):
    tol_l, tol_n, tol_f = tols
    assert set(losses) == set(ref_losses)
    e_loss = max(abs(v - ref_losses[k]) / max(1.0, abs(ref_losses[k])) for k, v in losses.items())
    named = dict(model.named_parameters())
    assert set(ref_norms) == set(named)
    e_norm, worst = 0.0, ""
    for k, p in named.items():
        e = abs(float(p.grad.double().norm()) - ref_norms[k]) / max(ref_norms[k], 1e-2)
        if e > e_norm:
            e_norm, worst = e, k
    e_full = {}
    for key, name, want in full:
        a, b = named[name].grad.detach().double().cpu().numpy(), want.astype(np.float64)
        e_full[key] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"\nengine vs reference, {tag}: loss err {e_loss:.2e}, worst grad-norm err {e_norm:.2e} ({worst}), full-gradient "
          f"Frobenius errs {({k: f'{v:.1e}' for k, v in e_full.items()})}", flush=True)
    assert e_loss <= tol_l, (e_loss, losses, ref_losses)
    assert e_norm <= tol_n, (e_norm, worst)
    assert max(e_full.values()) <= tol_f, e_full
    assert all(torch.isfinite(p.grad).all() for p in named.values())


def test_engine_step_matches_reference_pipeline_fixture(golden_dir):
    """g_pipeline_full.npz: two 16-beam frames, the full 6+2+2 model; the engine also prefetches a next batch inside the
    step (cross-step stage 1 on the decoder-B stream), as in the benchmark."""
    g = np.load(os.path.join(golden_dir, "g_pipeline_full.npz"))
    frames = [synth.lidar_frame(11, beams=16, n_az=400), synth.lidar_frame(12, beams=16, n_az=360)]
    model = _model()
    losses, tr, eng = _engine_step(model, frames, g["ids_keep"], g["ids_mask"], next_frames=frames)
    ref = dict(zip([str(n) for n in g["loss_names"]], [float(v) for v in g["loss_vals"]]))
    gn = dict(zip([str(n) for n in g["grad_names"]], [float(v) for v in g["grad_norms"]]))
    _compare("pipeline_full", model, losses, ref, gn, [(k, n, g[k]) for k, n in SMALL_FULL], TOL_SMALL)
    # which kernels this comparison pinned: every stack is small enough for the one-launch forward; the
    # encoder's backward is one launch per layer unless a window kept more than 64 pillars (an injected mask: unknown -> the
    # two-launch form), the decoders' keeps the two-launch form (dead rows, the mask-token sum)
    forms = eng.last_forms()
    assert forms["enc_fwd"] == forms["den_fwd"] == forms["cen_fwd"] == "one_launch", forms
    assert forms["den_bwd"] == forms["cen_bwd"] == "three_launch", forms
    assert forms["enc_bwd"] == ("one_launch" if eng.last_sizes()["big_bundle_layouts"] == 0 else "three_launch"), forms
    # the prefetched batch is pending and steps with a RANDOM mask next: finite, same sizes
    tr.flat.zero_grad()
    l2, _ = eng.step(eng.pending, None, 1e-5, run_optimizer=False)
    torch.cuda.synchronize()
    s = eng.last_sizes()
    assert torch.isfinite(l2).all() and s["V"] == int(g["ids_keep"].size + g["ids_mask"].size) and s["mask_draws"] == 2


def test_one_launch_layer_generic_body_matches_reference_fixture(golden_dir):
    """the 5-9-tile path of the one-launch layer (round 5: fused_fwd_body<9, false>; since round 6 the looping kernel of sst_ws.hip) runs in the
    product only for a window that kept more than 64 pillars, and was pinned by a self-comparison alone.  Here the one-launch
    form is forced for EVERY stack (geomae_sst_set_fused_layers(2)): the decoders' token sets pack bundles of up to 144
    positions, so their eight layers go through that body -- and the step is held to the reference fixture at the same
    bounds as the default path."""
    from geomae_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "g_pipeline_full.npz"))
    frames = [synth.lidar_frame(11, beams=16, n_az=400), synth.lidar_frame(12, beams=16, n_az=360)]
    ref = dict(zip([str(n) for n in g["loss_names"]], [float(v) for v in g["loss_vals"]]))
    gn = dict(zip([str(n) for n in g["grad_names"]], [float(v) for v in g["grad_norms"]]))
    try:
        lib.geomae_sst_set_fused_layers(2)
        model = _model()
        losses, tr, eng = _engine_step(model, frames, g["ids_keep"], g["ids_mask"])
        _compare("pipeline_full, one-launch layers in every stack", model, losses, ref, gn, [(k, n, g[k]) for k, n in SMALL_FULL], TOL_SMALL)
    finally:
        lib.geomae_sst_set_fused_layers(1)


@pytest.mark.parametrize("case", ["c2", "c3", "c4"])
def test_engine_step_matches_reference_at_full_size(golden_dir, case):
    """BASELINE configs 2 (the benchmarked one: 4 single-sweep frames), 3 (10 sweeps) and 4 (Waymo geometry)."""
    from fullsize_cases import CASES
    g = np.load(os.path.join(golden_dir, "g_fullsize.npz"))
    K = lambda k: g[f"{case}.{k}"]
    frames = [synth.lidar_frame(**kw) for kw in CASES[case][1]]
    assert [f.shape[0] for f in frames] == list(K("n_points"))
    model = _model(GEOM_WAYMO if case == "c4" else None)
    losses, tr, eng = _engine_step(model, frames, K("ids_keep"), K("ids_mask"))
    sz = eng.last_sizes()
    assert sz["V"] == int(K("V"))
    # which form the encoder took (round 5): the reference's mask is recounted by geomae_pretrain_set_mask; without a window that
    # kept more than 64 pillars the encoder's layers are ONE launch each, forward and backward -- at config 2 (the benchmarked
    # case, <= 12288 kept pillars) that is the form this comparison pins to the reference
    print(f"\n{case}: fullest windows {sz['max_window_keep']}, layouts with large bundles {sz['big_bundle_layouts']}", flush=True)
    if case == "c2":
        assert sz["big_bundle_layouts"] == 0 and max(sz["max_window_keep"]) <= 64, sz
    # ... and that it DID take it (the choice hangs on six run-time conditions; a silent fall-back to the
    # two-launch backward would keep this comparison green): the engine reports the form of every stack of the step
    forms = eng.last_forms()
    print(f"{case}: stack forms {forms}", flush=True)
    assert forms["den_fwd"] == forms["cen_fwd"] == "three_launch" and forms["den_bwd"] == forms["cen_bwd"] == "three_launch", forms
    want = "one_launch" if sz["n_keep"] <= 12288 else "three_launch"        # (GeomaeTuning.fused_max_tokens)
    assert forms["enc_fwd"] == want, (forms, sz)
    assert forms["enc_bwd"] == (want if sz["big_bundle_layouts"] == 0 else "three_launch"), (forms, sz)
    if case == "c2":
        assert forms["enc_fwd"] == "one_launch" and forms["enc_bwd"] == "one_launch", forms
    ref = dict(zip([str(n) for n in K("loss_names")], [float(v) for v in K("loss_vals")]))
    gn = dict(zip([str(n) for n in K("grad_names")], [float(v) for v in K("grad_norms")]))
    _compare(case, model, losses, ref, gn, [(k, n, K(k)) for k, n in BIG_FULL], TOL_BIG)


def test_engine_set_mask_rejects_bad_ids_and_mask_stream_survives_recreation():
    """set_mask validates the id count against the batch; the mask stream is a function of the step index, not of how
    often an engine was (re)created or a submission replaced (batches_drawn reset by _create)."""
    from geomae_amd.engine import PretrainEngine
    from geomae_amd.train import Trainer
    model = _model(enc=1, dec=1)
    tr = Trainer(model)
    pool = [[torch.as_tensor(synth.lidar_frame(600 + 10 * i + b, beams=16, n_az=400), device="cuda") for b in range(2)]
            for i in range(3)]
    eng = PretrainEngine(model, tr.flat, tr.opt, 10.0)
    eng.submit(pool[0])
    with pytest.raises(RuntimeError, match="pillars"):
        eng.set_mask(torch.arange(5), torch.arange(5, 9))
    ids = []
    for i in range(3):
        eng.step(pool[i], pool[i + 1] if i < 2 else None, 1e-5, run_optimizer=False)
        ids.append(eng.last_ids()[0].clone())
    torch.cuda.synchronize()
    # a second run: the engine is re-created before step 1 (as when the workspace grows), batch 1 is submitted twice
    eng2 = PretrainEngine(model, tr.flat, tr.opt, 10.0)
    eng2.step(pool[0], pool[1], 1e-5, run_optimizer=False)
    a0 = eng2.last_ids()[0].clone()
    eng2.max_pillars = 4 * eng2.max_pillars
    eng2._create(sum(p.shape[0] for p in pool[1]))
    eng2.submit(pool[1])
    eng2.submit(pool[1])                                     # a replaced submission burns nothing
    eng2.step(pool[1], pool[2], 1e-5, run_optimizer=False)
    a1 = eng2.last_ids()[0].clone()
    eng2.step(pool[2], None, 1e-5, run_optimizer=False)
    a2 = eng2.last_ids()[0].clone()
    torch.cuda.synchronize()
    for want, got in zip(ids, (a0, a1, a2)):
        assert torch.equal(want, got)
    assert eng2.last_sizes()["mask_draws"] == 3 and eng2.mask_draws == 3
