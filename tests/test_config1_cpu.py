"""BASELINE configs[0] (the reference's own CPU-runnable case) -- the oracle against the fixture the
REFERENCE produced at that geometry (tests/golden/g_pipeline_c1.npz, oracle/make_golden_fullsize.py c1): 0.5 m pillars
-> fp32 grid 205 (not a multiple of the 12-pillar window), sub-voxels 0.25 / 0.125 m, SST-tiny 1 + 1 blocks; the 16 k
uniform cloud, a LiDAR-ring cloud, both as one batch.  The GPU half is tests/test_gpu_config1.py."""
import os

import numpy as np
import pytest
import torch

import geomae_oracle as O
from fullsize_cases import C1, CASES_C1, make_frames


def oracle_cfg():
    return O.mae_sst_cfg(*C1["blocks"], voxel=C1["top"], low=C1["low"], med=C1["med"], pc_range=C1["range"], grid=C1["grid"])


@pytest.mark.parametrize("case", list(CASES_C1))
def test_oracle_matches_reference_at_config1_geometry(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "g_pipeline_c1.npz"))
    K = lambda k: g[f"{case}.{k}"]
    frames = make_frames(CASES_C1[case][1])
    assert [f.shape[0] for f in frames] == list(K("n_points"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    params = {k: v.clone().requires_grad_(True) for k, v in O.make_params(7, *C1["blocks"]).items()}
    losses, aux = O.forward_train(params, frames, oracle_cfg(), ids_keep=K("ids_keep").astype(np.int64),
                                  ids_mask=K("ids_mask").astype(np.int64))
    assert np.array_equal(aux["voxel_coors"], K("voxel_coors").astype(aux["voxel_coors"].dtype))      # order of torch.unique
    assert aux["voxel_coors"][:, 2:].max() <= 204
    ref = dict(zip([str(n) for n in K("loss_names")], K("loss_vals")))
    for k, v in losses.items():
        assert abs(float(v) - ref[k]) <= 2e-4 * max(1.0, abs(ref[k])), (k, float(v), ref[k])
    sum(losses.values()).backward()
    gn = dict(zip([str(n) for n in K("grad_names")], K("grad_norms")))
    for k, p in params.items():
        if k in gn:
            assert abs(float(p.grad.double().norm()) - gn[k]) <= 2e-3 * max(gn[k], 1e-2), (k, float(p.grad.norm()), gn[k])
    a = params["voxel_encoder.vfe_layers.0.linear.weight"].grad.double().numpy()
    b = K("grad_vfe0").astype(np.float64)
    assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b)
