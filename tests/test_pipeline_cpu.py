"""Host side of the GPU input pipeline (N2): config parsing, the order of the random draws, the oracle restatement."""
import os
import sys

import numpy as np
import pytest

import geomae_oracle as O
from geomae_amd.pipeline import GpuTrainPipeline

RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def make_frames(seed, n_frames=2, n_sweeps=(3, 0), n_pts=4000):
    rng = np.random.default_rng(seed)
    frames = []
    for b in range(n_frames):
        def cloud(n):
            p = np.empty((n, 5), np.float32)
            p[:, :2] = rng.uniform(-60, 60, (n, 2))
            p[: n // 10, :2] = rng.uniform(-1.5, 1.5, (n // 10, 2))       # some points inside the remove_close box
            p[:, 2] = rng.uniform(-6, 4, n)
            p[:, 3] = rng.uniform(0, 255, n)
            p[:, 4] = rng.uniform(0, 1, n)
            return p
        sweeps = []
        for k in range(n_sweeps[b % len(n_sweeps)]):
            a = rng.uniform(-0.05, 0.05)
            R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float64)
            sweeps.append(dict(points=cloud(n_pts // 2 + 17 * k), sensor2lidar_rotation=R,
                               sensor2lidar_translation=rng.uniform(-1, 1, 3), timestamp=1_600_000_000_000_000 - 50_000 * (k + 1)))
        frames.append(dict(points=cloud(n_pts), timestamp=1_600_000_000.0, sweeps=sweeps))
    return frames


def test_from_reference_config_and_draw_order():
    cfg = [dict(type="LoadPointsFromFile", coord_type="LIDAR", load_dim=5, use_dim=5),
           dict(type="LoadPointsFromMultiSweeps", sweeps_num=9, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True, remove_close=True),
           dict(type="GlobalRotScaleTrans", rot_range=[-0.3925, 0.3925], scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0]),
           dict(type="RandomFlip3D", sync_2d=False, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
           dict(type="PointsRangeFilter", point_cloud_range=RANGE), dict(type="PointShuffle"),
           dict(type="DefaultFormatBundle3D", class_names=[]), dict(type="Collect3D", keys=["points"])]
    pipe = GpuTrainPipeline.from_config(cfg)
    assert pipe.sweeps_num == 9 and pipe.remove_close and pipe.pad_empty_sweeps and pipe.shuffle
    assert pipe.point_cloud_range == RANGE and pipe.rot_range == (-0.3925, 0.3925)
    # the draws follow the reference's order of np.random calls
    fr = dict(points=np.zeros((1, 5), np.float32), timestamp=0.0, sweeps=[dict()] * 12)
    rs = np.random.RandomState(7)
    d = pipe.draw(fr, rs)
    ref = np.random.RandomState(7)
    choices = ref.choice(12, 9, replace=False)
    rot, scale = ref.uniform(-0.3925, 0.3925), ref.uniform(0.95, 1.05)
    ref.normal(scale=[0, 0, 0], size=3)
    ref.choice(2)
    fh, fv = ref.rand() < 0.5, ref.rand() < 0.5
    assert d.sweep_choices == list(choices) and d.rotation == rot and d.scale == scale
    assert (d.flip_horizontal, d.flip_vertical) == (fh, fv) and d.shuffle_seed != 0


def test_oracle_pipeline_semantics():
    frames = make_frames(1)
    pipe = GpuTrainPipeline(RANGE, sweeps_num=2)
    d = pipe.draw(frames[0], np.random.RandomState(3))
    out = O.train_pipeline_cpu(frames[0], d, RANGE, sweeps_num=2)
    assert out.dtype == np.float32 and out.shape[1] == 5
    assert (out[:, 0] > RANGE[0]).all() and (out[:, 0] < RANGE[3]).all() and (out[:, 2] > RANGE[2]).all()
    lags = np.unique(out[:, 4])
    assert lags[0] == 0.0 and len(lags) == 3 and np.allclose(sorted(lags)[1:], [0.05, 0.1], atol=1e-6)
    # empty sweeps: the key frame repeated sweeps_num times with the close box removed from the copies
    d1 = pipe.draw(frames[1], np.random.RandomState(4))
    out1 = O.train_pipeline_cpu(frames[1], d1, RANGE, sweeps_num=2)
    assert (out1[:, 4] == 0).all() and out1.shape[0] > frames[1]["points"].shape[0]
