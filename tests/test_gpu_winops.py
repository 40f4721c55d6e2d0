"""Operator-level window API (mmdet3d/ops/__init__.py:22-26) on the HIP library vs the reference's own
mmdet3d/ops/sst/sst_ops.py (tests/golden/g_winops.npz, oracle/make_golden_winops.py), plus Voxelization_with_flag."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))


def _inputs():
    from make_golden_winops import DROP, inputs          # the seeded generator only (no reference import at module level)
    win, lvl, feat = inputs()
    return DROP, torch.as_tensor(win).cuda(), torch.as_tensor(lvl).cuda(), torch.as_tensor(feat).cuda()


def test_window_operators_match_reference_fixture(golden_dir):
    from geomae_amd import ops
    g = np.load(os.path.join(golden_dir, "g_winops.npz"))
    DROP, win, lvl, feat = _inputs()
    assert win.shape[0] == int(g["n"])
    conti = ops.make_continuous_inds(win)
    assert conti.dtype == win.dtype and np.array_equal(conti.cpu().numpy(), g["conti_all"])
    inner = ops.get_inner_win_inds(win).cpu().numpy()
    w = win.cpu().numpy()
    for k, wid in enumerate(np.unique(w)):                       # a permutation of 0..M-1 inside every window
        r = np.sort(inner[w == wid])
        assert np.array_equal(r, np.arange(r.shape[0])) and r[-1] == g["inner_max_per_window"][k]
    inds = ops.get_flat2win_inds(win, lvl, DROP, debug=True)
    x = feat.clone().requires_grad_(True)
    f3d = ops.flat2window(x, lvl, inds, DROP)
    assert sorted(f3d) == [0, 1, 2]
    for dl in DROP:
        flat2win, where = inds[dl]
        T = DROP[dl]["max_tokens"]
        assert np.array_equal(where[0].cpu().numpy(), g[f"l{dl}.where"])
        assert np.array_equal((flat2win // T).cpu().numpy(), g[f"l{dl}.window_of_token"])
        t = f3d[dl]
        assert list(t.shape) == list(g[f"l{dl}.shape"])
        got = np.sort(t.detach().double().sum(-1).cpu().numpy(), axis=1)
        assert np.allclose(got, g[f"l{dl}.sorted_rowsums"], rtol=0, atol=1e-12)          # same rows per window, zero padding
        assert np.array_equal((t.detach().abs().sum(-1) > 0).sum(1).cpu().numpy(), g[f"l{dl}.nonzero_rows"])
    back = ops.window2flat(f3d, inds)
    assert torch.equal(back.detach(), feat)
    # differentiable like the reference's index copies: d/dfeat of sum(window2flat(2 * flat2window(feat)) * w) = 2 w
    wgt = torch.randn(feat.shape, generator=torch.Generator().manual_seed(1)).cuda()
    (ops.window2flat({k: 2 * v for k, v in f3d.items()}, inds) * wgt).sum().backward()
    assert torch.allclose(x.grad, 2 * wgt)


def test_rows_copy_handles_odd_rows_and_empty_input():
    from geomae_amd import ops
    src = torch.arange(7 * 3, dtype=torch.float32, device="cuda").reshape(7, 3)           # 12-byte rows (4-byte path)
    idx = torch.tensor([4, 0, 6], device="cuda")
    got = ops._RowsCopy.apply(src, idx, 3, False)
    assert torch.equal(got, src[idx])
    sc = ops._RowsCopy.apply(got, idx, 7, True)
    want = torch.zeros_like(src)
    want[idx] = src[idx]
    assert torch.equal(sc, want)
    b = torch.arange(5 * 3, dtype=torch.uint8, device="cuda").reshape(5, 3)               # 3-byte rows (byte path)
    assert torch.equal(ops._RowsCopy.apply(b, torch.tensor([3, 1], device="cuda"), 2, False), b[[3, 1]])
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    assert ops.make_continuous_inds(e).numel() == 0 and ops.get_inner_win_inds(e).numel() == 0
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.make_continuous_inds(torch.zeros(4, dtype=torch.int64))


def test_voxelization_with_flag_returns_the_reference_tuple():
    """ops/voxel/voxelize.py:126-244: (voxels, voxels_flag, coors, num_points_per_voxel); flag = occupied slots."""
    from geomae_amd import ops, synth
    pts = torch.as_tensor(synth.lidar_frame(5, beams=16, n_az=600), device="cuda")
    kw = dict(voxel_size=(0.25, 0.25, 8.0), point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], max_num_points=20,
              max_voxels=30000)
    plain = ops.Voxelization(**kw)(pts)
    voxels, flag, coors, num = ops.Voxelization_with_flag(**kw)(pts)
    assert torch.equal(voxels, plain[0]) and torch.equal(coors, plain[1]) and torch.equal(num, plain[2])
    assert flag.dtype == torch.bool and flag.shape == voxels.shape[:2]
    assert torch.equal(flag.sum(1).int(), num)
    assert torch.equal(flag, torch.arange(20, device="cuda")[None, :] < num[:, None])
    # every flagged slot holds a point; the rest of the voxel tensor is zero
    assert float(voxels[~flag].abs().sum()) == 0.0
    dyn = ops.Voxelization_with_flag(voxel_size=(0.25, 0.25, 8.0), point_cloud_range=kw["point_cloud_range"],
                                     max_num_points=-1, max_voxels=(-1, -1))(pts)
    assert dyn.shape == (pts.shape[0], 3)
