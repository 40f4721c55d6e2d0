"""GPU parity of the one-launch SST layer (csrc/sst_fused.hip) against the three-launch form (qkv / attention / ffn
kernels) it replaces inside geomae_sst_stack_forward: same inputs, same packed weights, same window layouts."""
import numpy as np
import pytest
import torch

import geomae_oracle as O
from geomae_amd import synth

pytestmark = pytest.mark.gpu

RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
TOP = (0.256, 0.256, 8)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from geomae_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _model(dev, enc, dec):
    import geomae_amd
    from geomae_amd.configs import mae_sst_model
    cfg = mae_sst_model(encoder_num_blocks=enc, decoder_num_blocks=dec)
    cfg["backbone"]["compute_dtype"] = "bf16"
    model = geomae_amd.build_model(cfg).to(dev)
    model.load_state_dict(O.make_params(7, enc, dec), strict=False)
    return model.train()


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("case", ["encoder_small", "decoder_tail", "gather_rows"])
def test_one_launch_layer_matches_three_launch_form(dev, case):
    """Forward output, and -- through every saved activation -- the input gradient and all parameter gradients of the
    (unfused) backward, of a training stack run both ways.  The two forms differ in rounding only (LayerNorm statistics
    merged from per-wave partial moments; V projected in two orientations), so the bounds are those of bf16 noise.
    encoder_small: a third of the pillars (small windows, bundles of <= 4 tiles); decoder_tail: all pillars with a fill
    row for the last third (windows of up to 144 tokens: the 9-tile body); gather_rows: the stack's input row map."""
    from geomae_amd import ops, _lib
    lib = _lib.load()
    model = _model(dev, 2, 2)
    bb = model.backbone
    frames = [synth.lidar_frame(71), synth.lidar_frame(72, beams=24, n_az=700), synth.lidar_frame(73, beams=16, n_az=300)]
    _, coors = O.voxelize_batch(frames, TOP, RANGE)
    vc = O.unique_rows(coors)[0]
    gen = torch.Generator().manual_seed(9)
    if case == "encoder_small":
        keep = np.sort(np.random.default_rng(3).permutation(vc.shape[0])[: vc.shape[0] // 3])
        vc = vc[keep]
    vc = torch.as_tensor(vc, device=dev)
    n = vc.shape[0]
    name = "enc" if case == "encoder_small" else "cen"
    blocks = bb.encoder_blocks if name == "enc" else bb.decoder_centroid_blocks
    nl = 2 * len(blocks)
    bb._packed.refresh()
    layouts, _ = bb.get_voxel_info(vc, len(frames))
    assert layouts[0].fbun_tok is not None
    nb = int(layouts[0].num_fbundles.item())
    sizes = (layouts[0].fbun_tok[1:nb + 1] - layouts[0].fbun_tok[:nb]).cpu().numpy()
    assert sizes.min() >= 1 and sizes.max() <= 144
    if case == "decoder_tail":
        assert sizes.max() > 64, "the case is meant to reach the 9-tile body"
    w = bb._packed.weight_array(bb._stack_base[name], nl)
    kw = {}
    if case == "decoder_tail":
        n_in = n - n // 3 - 5
        x = torch.randn(n_in, 128, generator=gen).to(dev)
        kw["tail"] = (torch.randn(1, 128, generator=gen).to(dev), n - n_in)
    elif case == "gather_rows":
        V = n + n // 2 + 3
        x = torch.randn(V, 128, generator=gen).to(dev)
        kw["rows"] = torch.randperm(V, generator=gen)[:n].int().to(dev)
    else:
        x = torch.randn(n, 128, generator=gen).to(dev)
    dz = torch.randn(n, 128, generator=gen).to(dev)
    res = []
    try:
        for mode in (0, 2):
            lib.geomae_sst_set_fused_layers(mode)
            for p in bb.parameters():
                p.grad = None
            g = bb._packed.grad_array(bb._stack_base[name], nl)
            z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0], **kw)
            dx = ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
            torch.cuda.synchronize()
            res.append((z.clone(), dx.clone(), {k: v.grad.clone() for k, v in blocks.named_parameters()}))
    finally:
        lib.geomae_sst_set_fused_layers(1)
    (z0, d0, g0), (z1, d1, g1) = res
    assert torch.isfinite(z1).all() and torch.isfinite(d1).all()
    assert _rel(z1, z0) < 4e-3, _rel(z1, z0)
    assert _rel(d1, d0) < 1e-2, _rel(d1, d0)
    bad = {k: _rel(g1[k], g0[k]) for k in g0 if _rel(g1[k], g0[k]) > 2e-2}
    assert not bad, bad


@pytest.mark.parametrize("case", ["decoder_tail", "gather_rows", "encoder_small", "decoder_plain", "cap_144"])
def test_weight_stationary_layer_matches_three_launch_form(dev, case):
    """csrc/sst_ws.hip (one launch per layer, workgroups loop over bundles with the layer's weights in registers; windows of up
    to 144 positions through the run-time tile loops and the online softmax) against the three-launch form inside
    geomae_sst_stack_forward / _backward: forward output and -- through every saved activation -- input gradient and all
    parameter gradients.  decoder_tail: all pillars, fill row for the last third (9-tile windows); gather_rows: the input
    row map; encoder_small: small windows only; decoder_plain: plain input; cap_144: bundles packed up to 144 positions
    (several windows per 9-tile bundle: the block-diagonal mask across tile pairs)."""
    from geomae_amd import ops, _lib
    lib = _lib.load()
    model = _model(dev, 2, 2)
    bb = model.backbone
    frames = [synth.lidar_frame(71), synth.lidar_frame(72, beams=24, n_az=700), synth.lidar_frame(73, beams=16, n_az=300)]
    _, coors = O.voxelize_batch(frames, TOP, RANGE)
    vc = O.unique_rows(coors)[0]
    gen = torch.Generator().manual_seed(11)
    if case == "encoder_small":
        keep = np.sort(np.random.default_rng(3).permutation(vc.shape[0])[: vc.shape[0] // 3])
        vc = vc[keep]
    vc = torch.as_tensor(vc, device=dev)
    n = vc.shape[0]
    name = "enc" if case == "encoder_small" else "cen"
    blocks = bb.encoder_blocks if name == "enc" else bb.decoder_centroid_blocks
    nl = 2 * len(blocks)
    bb._packed.refresh()
    old = _lib.set_tuning(ws_layers=2, fused_layers=1, bundle_cap=144 if case == "cap_144" else 0)
    try:
        layouts, _ = bb.get_voxel_info(vc, len(frames))
        nb = int(layouts[0].num_fbundles.item())
        sizes = (layouts[0].fbun_tok[1:nb + 1] - layouts[0].fbun_tok[:nb]).cpu().numpy()
        assert sizes.min() >= 1 and sizes.max() <= 144
        if case in ("decoder_tail", "decoder_plain", "cap_144"):
            assert sizes.max() > 64, "the case is meant to reach windows of more than four tiles"
        w = bb._packed.weight_array(bb._stack_base[name], nl)
        kw = {}
        if case == "decoder_tail":
            n_in = n - n // 3 - 5
            x = torch.randn(n_in, 128, generator=gen).to(dev)
            kw["tail"] = (torch.randn(1, 128, generator=gen).to(dev), n - n_in)
        elif case == "gather_rows":
            V = n + n // 2 + 3
            x = torch.randn(V, 128, generator=gen).to(dev)
            kw["rows"] = torch.randperm(V, generator=gen)[:n].int().to(dev)
        else:
            x = torch.randn(n, 128, generator=gen).to(dev)
        dz = torch.randn(n, 128, generator=gen).to(dev)
        res = []
        for ws in (0, 2):
            _lib.set_tuning(ws_layers=ws, fused_layers=0 if ws == 0 else 1)
            for p in bb.parameters():
                p.grad = None
            g = bb._packed.grad_array(bb._stack_base[name], nl)
            z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0], **kw)
            dx = ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved)
            torch.cuda.synchronize()
            res.append((z.clone(), dx.clone(), {k: v.grad.clone() for k, v in blocks.named_parameters()}))
    finally:
        _lib.set_tuning(**old)
    (z0, d0, g0), (z1, d1, g1) = res
    assert torch.isfinite(z1).all() and torch.isfinite(d1).all()
    assert _rel(z1, z0) < 4e-3, _rel(z1, z0)
    assert _rel(d1, d0) < 1e-2, _rel(d1, d0)
    bad = {k: _rel(g1[k], g0[k]) for k in g0 if _rel(g1[k], g0[k]) > 2e-2}
    assert not bad, bad


def _encoder_small_case(dev, seed=9):
    from geomae_amd import ops
    model = _model(dev, 2, 2)
    bb = model.backbone
    frames = [synth.lidar_frame(71), synth.lidar_frame(72, beams=24, n_az=700), synth.lidar_frame(73, beams=16, n_az=300)]
    _, coors = O.voxelize_batch(frames, TOP, RANGE)
    vc = O.unique_rows(coors)[0]
    keep = np.sort(np.random.default_rng(3).permutation(vc.shape[0])[: vc.shape[0] // 3])
    vc = torch.as_tensor(vc[keep], device=dev)
    bb._packed.refresh()
    layouts, _ = bb.get_voxel_info(vc, len(frames))
    gen = torch.Generator().manual_seed(seed)
    n = vc.shape[0]
    return model, bb, layouts, n, torch.randn(n, 128, generator=gen).to(dev), torch.randn(n, 128, generator=gen).to(dev)


def test_one_launch_backward_matches_two_launch_backward_at_stack_level(dev):
    """sst_layer_bwd_kernel pinned DIRECTLY -- geomae_sst_stack_backward with every contraction queued
    (defer_last_weight_grad = 2) and the promise "no bundle of more than four tiles", once with GeomaeTuning.fused_bwd on and
    once off, on the same saved activations of a one-launch forward.  geomae_sst_last_stack_forms says which backward ran (a
    silent fall-back to the two-launch form would make this a self-comparison), geomae_sst_fused_dropped_bundles that no bundle
    was skipped."""
    from geomae_amd import ops, _lib
    model, bb, layouts, n, x, dz = _encoder_small_case(dev)
    blocks = bb.encoder_blocks
    nl = 2 * len(blocks)
    for L in layouts:
        nb = int(L.num_fbundles.item())
        sizes = (L.fbun_tok[1:nb + 1] - L.fbun_tok[:nb]).cpu().numpy()
        assert sizes.max() <= 64, "the case must hold no bundle of more than four tiles"
    w = bb._packed.weight_array(bb._stack_base["enc"], nl)
    ops.fused_dropped_bundles(reset=True)
    res, forms = [], []
    old = _lib.get_tuning().fused_bwd
    try:
        for fused_bwd in (0, 1):
            _lib.set_tuning(fused_bwd=fused_bwd)
            for p in bb.parameters():
                p.grad = None
            g = bb._packed.grad_array(bb._stack_base["enc"], nl)
            z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0], big_layouts=0)
            dx, scratch = ops.sst_stack_backward(dz, n, w, g, layouts, bb.pos_table, bb.nhead[0], saved, defer_all=True, big_layouts=0)
            forms.append(ops.last_stack_forms())
            ops.flush_weight_grad()
            torch.cuda.synchronize()
            res.append((z.clone(), dx.clone(), {k: v.grad.clone() for k, v in blocks.named_parameters()}))
            del scratch
    finally:
        _lib.set_tuning(fused_bwd=old)
    assert forms == [(1, 0), (1, 1)], forms                     # one-launch forward both times; two-launch, then one-launch backward
    assert ops.fused_dropped_bundles() == 0
    (z0, d0, g0), (z1, d1, g1) = res
    assert torch.equal(z0, z1)
    assert torch.isfinite(d1).all()
    assert _rel(d1, d0) < 1e-2, _rel(d1, d0)
    bad = {k: _rel(g1[k], g0[k]) for k in g0 if _rel(g1[k], g0[k]) > 2e-2}
    assert not bad, bad


def test_a_broken_big_bundle_promise_is_counted(dev):
    """the one-launch kernels skip bundles of more than four tiles on a host-side promise.  A wrong promise (here: the
    decoders' token set, windows of up to 144 pillars, declared free of large bundles) must not pass silently: the kernels count
    every bundle no launch ran, geomae_sst_fused_dropped_bundles reads the count."""
    from geomae_amd import ops, _lib
    lib = _lib.load()
    model = _model(dev, 2, 2)
    bb = model.backbone
    frames = [synth.lidar_frame(71), synth.lidar_frame(72, beams=24, n_az=700)]
    _, coors = O.voxelize_batch(frames, TOP, RANGE)
    vc = torch.as_tensor(O.unique_rows(coors)[0], device=dev)
    n = vc.shape[0]
    bb._packed.refresh()
    old = _lib.set_tuning(bundle_cap=48)
    try:
        layouts, _ = bb.get_voxel_info(vc, len(frames))
        nb = int(layouts[0].num_fbundles.item())
        sizes = (layouts[0].fbun_tok[1:nb + 1] - layouts[0].fbun_tok[:nb]).cpu().numpy()
        n_big = int((sizes > 64).sum())
        assert n_big > 0
        nl = 2 * len(bb.decoder_centroid_blocks)
        w = bb._packed.weight_array(bb._stack_base["cen"], nl)
        x = torch.randn(n, 128, device=dev)
        ops.fused_dropped_bundles(reset=True)
        lib.geomae_sst_set_fused_layers(2)
        ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])                  # no promise: the second launch runs
        assert ops.fused_dropped_bundles() == 0
        ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0], big_layouts=0)   # a wrong promise
        assert ops.fused_dropped_bundles() >= n_big
    finally:
        lib.geomae_sst_set_fused_layers(1)
        _lib.set_tuning(**old)


def test_forward_work_items_cover_every_tile_once_and_match_the_unsplit_forward(dev):
    """Round 6: the one-launch forward walks WORK ITEMS (window.hip item_pack: a 32-position packing whose bundles of three / four
    tiles are split by query tile into two items, all in one round of workgroups).  The item list must own every tile of every
    bundle exactly once; the forward through it must equal the forward over the second packing (GeomaeTuning.fwd_item_cap = 0)
    up to the summation order of the softmax (another tile partition of the same windows)."""
    from geomae_amd import ops, _lib
    old_cap = _lib.set_tuning(fwd_item_cap=32)["fwd_item_cap"]          # (off by default: the layouts below are built with it on)
    try:
        model, bb, layouts, n, x, dz = _encoder_small_case(dev)
    finally:
        _lib.set_tuning(fwd_item_cap=old_cap)
    nl = 2 * len(bb.encoder_blocks)
    w = bb._packed.weight_array(bb._stack_base["enc"], nl)
    n_split = 0
    for L in layouts:
        ni = int(L.num_fitems.item())
        assert 0 < ni <= 2 * (L.max_windows + 1)
        it = L.fitems[:ni].cpu().numpy()
        owned = {}
        for s0, T, q0, nq in it:
            nt = (T + 15) // 16
            assert 1 <= T <= 144 and 0 <= q0 and nq >= 1 and q0 + nq <= nt, (s0, T, q0, nq)
            n_split += int(nq < nt)
            for q in range(q0, q0 + nq):
                assert (s0, q) not in owned
                owned[(s0, q)] = 1
        # whole bundles: every tile of every bundle start that appears is owned, and the bundles tile the position range
        starts = sorted({(int(a), int(b)) for a, b, _, _ in it})
        assert starts[0][0] == 0 and all(starts[i][0] + starts[i][1] == starts[i + 1][0] for i in range(len(starts) - 1))
        assert starts[-1][0] + starts[-1][1] == n
        assert len(owned) == sum((T + 15) // 16 for _, T in starts)
    assert n_split > 0, "the case is meant to hold bundles of three tiles"
    res = []
    old = _lib.get_tuning().fwd_item_cap
    try:
        for cap in (0, 32):
            _lib.set_tuning(fwd_item_cap=cap)
            ops.fused_dropped_bundles(reset=True)
            z, saved = ops.sst_stack_forward(x, w, layouts, bb.pos_table, bb.nhead[0])
            torch.cuda.synchronize()
            assert ops.last_stack_forms()[0] == 1 and ops.fused_dropped_bundles() == 0
            res.append((z.clone(), saved.clone()))
    finally:
        _lib.set_tuning(fwd_item_cap=old)
    assert torch.isfinite(res[1][0]).all()
    assert _rel(res[1][0], res[0][0]) < 2e-3, _rel(res[1][0], res[0][0])
