"""GPU input pipeline for the pre-training loop.

The reference prepares every sample on CPU workers (4 per GPU): read the key frame and up to 9 earlier sweeps,
remove close points, move each sweep into the key frame's lidar frame, set the time lag, concatenate
(LoadPointsFromMultiSweeps, mmdet3d/datasets/pipelines/loading.py:184-233), then GlobalRotScaleTrans, RandomFlip3D,
PointsRangeFilter and PointShuffle (transforms_3d.py:734-757, 125-160, 849-884, 770-797), configured by
configs/mae_sst/...6x_1e-5.py:167-197.  At 8 GPUs x 4 frames x ~270 k points that is ~170 MB of fp32 per step
going through numpy; here the files are still read on the host (np.fromfile is I/O), but everything per point runs
in three launches per BATCH (csrc/points_pipeline.hip) on the raw buffers, and the result is already the
concatenated `points` + offsets the voxelizer wants.

Only the handful of random scalars per frame are drawn on the host, in the reference's order, from numpy's global
stream by default, so a seeded run reproduces the reference's augmentation parameters.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import GeomaeFrameAug, GeomaeSweepInfo, check, f3


class FrameDraw:
    """The random decisions of one sample, in the order the reference pipeline makes them."""
    __slots__ = ("sweep_choices", "rotation", "scale", "translation", "flip_horizontal", "flip_vertical", "shuffle_seed")


class GpuTrainPipeline:
    def __init__(self, point_cloud_range, sweeps_num=9, remove_close=True, pad_empty_sweeps=True, test_mode=False,
                 rot_range=(-0.3925, 0.3925), scale_ratio_range=(0.95, 1.05), translation_std=(0, 0, 0),
                 flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5, shuffle=True, close_radius=1.0,
                 mmdet_flip_draw=True):
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.sweeps_num, self.remove_close, self.pad_empty_sweeps, self.test_mode = sweeps_num, remove_close, pad_empty_sweeps, test_mode
        self.rot_range, self.scale_ratio_range = tuple(rot_range), tuple(scale_ratio_range)
        self.translation_std = [float(v) for v in translation_std]
        self.flip_h, self.flip_v, self.shuffle, self.close_radius = flip_ratio_bev_horizontal, flip_ratio_bev_vertical, shuffle, close_radius
        # mmdet's RandomFlip.__call__ (the base class, un-vendored) consumes one np.random.choice before RandomFlip3D
        # draws its own two numbers (transforms_3d.py:138); kept so that a seeded run matches the reference's stream
        self.mmdet_flip_draw = mmdet_flip_draw

    @classmethod
    def from_config(cls, train_pipeline, point_cloud_range=None):
        """Build from the `train_pipeline` list of a reference config (configs/mae_sst/*.py:167-197)."""
        kw = {}
        for t in train_pipeline:
            ty = t["type"]
            if ty == "LoadPointsFromMultiSweeps":
                kw.update(sweeps_num=t.get("sweeps_num", 10), remove_close=t.get("remove_close", False),
                          pad_empty_sweeps=t.get("pad_empty_sweeps", False), test_mode=t.get("test_mode", False))
            elif ty == "GlobalRotScaleTrans":
                kw.update(rot_range=t.get("rot_range", [-0.78539816, 0.78539816]),
                          scale_ratio_range=t.get("scale_ratio_range", [0.95, 1.05]),
                          translation_std=t.get("translation_std", [0, 0, 0]))
            elif ty == "RandomFlip3D":
                kw.update(flip_ratio_bev_horizontal=t.get("flip_ratio_bev_horizontal", 0.0),
                          flip_ratio_bev_vertical=t.get("flip_ratio_bev_vertical", 0.0))
            elif ty == "PointsRangeFilter":
                point_cloud_range = t["point_cloud_range"]
            elif ty == "PointShuffle":
                kw["shuffle"] = True
        kw.setdefault("shuffle", False)
        return cls(point_cloud_range, **kw)

    # ------------------------------------------------------------------ host: random draws (reference order)
    def draw(self, frame, rng=np.random):
        d = FrameDraw()
        n_sw = len(frame.get("sweeps", []))
        if n_sw <= self.sweeps_num:
            d.sweep_choices = list(range(n_sw))
        elif self.test_mode:
            d.sweep_choices = list(range(self.sweeps_num))
        else:
            d.sweep_choices = [int(v) for v in rng.choice(n_sw, self.sweeps_num, replace=False)]     # loading.py:211-212
        d.rotation = float(rng.uniform(self.rot_range[0], self.rot_range[1]))                          # :682
        d.scale = float(rng.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1]))             # :730
        d.translation = [float(v) for v in np.atleast_1d(rng.normal(scale=self.translation_std, size=3))]   # :663
        if self.mmdet_flip_draw:
            # mmdet 2.20 RandomFlip.__call__: np.random.choice([direction, None], p=[ratio, 1 - ratio]) -- with p the
            # draw is one uniform double (a plain choice(2) would consume the stream differently; pinned by
            # tests/golden/g_input_pipeline.npz, where the following two rand() calls are the reference's)
            rng.choice(2, p=[self.flip_h, 1.0 - self.flip_h])
        d.flip_horizontal = bool(rng.rand() < self.flip_h)                                             # :144
        d.flip_vertical = bool(rng.rand() < self.flip_v)                                               # :148
        d.shuffle_seed = int(rng.randint(1, 2 ** 31 - 1)) * (2 ** 31) + int(rng.randint(1, 2 ** 31 - 1)) if self.shuffle else 0
        return d

    # ------------------------------------------------------------------ host: gather raw buffers + descriptors
    def assemble(self, frames, draws):
        chunks, sweep_offsets, frame_offsets, infos, augs = [], [0], [0], [], []
        for b, (fr, d) in enumerate(zip(frames, draws)):
            key = np.ascontiguousarray(fr["points"], dtype=np.float32)

            def add(points, info):
                chunks.append(points)
                sweep_offsets.append(sweep_offsets[-1] + points.shape[0])
                infos.append(info)

            def info(rot=None, trans=None, dt=0.0, remove_close=False):
                s = GeomaeSweepInfo()
                s.rot[:] = list(np.asarray(rot if rot is not None else np.eye(3), dtype=np.float64).reshape(-1))
                s.trans[:] = list(np.asarray(trans if trans is not None else np.zeros(3), dtype=np.float64).reshape(-1))
                s.dt, s.frame, s.remove_close, s.has_transform = float(np.float32(dt)), b, int(remove_close), int(rot is not None)
                return s
            add(key, info())                                                        # key frame: dt := 0 (loading.py:199)
            sweeps = fr.get("sweeps", [])
            if self.pad_empty_sweeps and len(sweeps) == 0:
                for _ in range(self.sweeps_num):                                    # :202-207: copies of the key frame
                    add(key, info(remove_close=self.remove_close))
            else:
                ts = fr["timestamp"]
                for idx in d.sweep_choices:
                    sw = sweeps[idx]
                    add(np.ascontiguousarray(sw["points"], dtype=np.float32).reshape(-1, key.shape[1]),
                        info(sw["sensor2lidar_rotation"], sw["sensor2lidar_translation"], ts - sw["timestamp"] / 1e6,
                             self.remove_close))
            frame_offsets.append(sweep_offsets[-1])
            a = GeomaeFrameAug()
            rot32 = np.float32(d.rotation)                     # points.rotate: tensor.new_tensor(angle) -> fp32 sin / cos
            a.rot_cos, a.rot_sin, a.scale = float(np.cos(rot32)), float(np.sin(rot32)), float(np.float32(d.scale))
            a.trans[:] = [float(np.float32(v)) for v in d.translation]
            a.flip_horizontal, a.flip_vertical = int(d.flip_horizontal), int(d.flip_vertical)
            a.shuffle_seed_lo, a.shuffle_seed_hi = d.shuffle_seed & 0xFFFFFFFF, (d.shuffle_seed >> 32) & 0xFFFFFFFF
            augs.append(a)
        raw = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 5), np.float32)
        return raw, np.asarray(sweep_offsets, np.int32), np.asarray(frame_offsets, np.int32), infos, augs

    # ------------------------------------------------------------------ device
    def __call__(self, frames, device, draws=None, rng=np.random):
        """frames: list of dict(points [N,5] f32, timestamp (s), sweeps=[dict(points, sensor2lidar_rotation [3,3],
        sensor2lidar_translation [3], timestamp (us))]).  -> list of [N_b, 5] tensors (views of one buffer)."""
        from .ops import _ptr, _stream
        if draws is None:
            draws = [self.draw(fr, rng) for fr in frames]
        raw, sweep_off, frame_off, infos, augs = self.assemble(frames, draws)
        lib = _lib.load()

        def to_dev(arr):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            return t.pin_memory().to(device, non_blocking=True)

        def structs_to_dev(items, ty):
            buf = (ty * len(items))(*items)
            return to_dev(np.frombuffer(buf, dtype=np.uint8).copy())
        n, nf = raw.shape
        d_raw, d_so, d_fo = to_dev(raw), to_dev(sweep_off), to_dev(frame_off)
        d_inf, d_aug = structs_to_dev(infos, GeomaeSweepInfo), structs_to_dev(augs, GeomaeFrameAug)
        out = torch.empty((max(n, 1), nf), dtype=torch.float32, device=device)
        out_off = torch.empty(len(frames) + 1, dtype=torch.int32, device=device)
        wsb = lib.geomae_points_pipeline_workspace_bytes(n, nf)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=device)
        check(lib.geomae_points_pipeline(_ptr(d_raw), n, nf, _ptr(d_so), _ptr(d_inf), len(infos), _ptr(d_fo), _ptr(d_aug),
                                         len(frames), f3(self.point_cloud_range), float(self.close_radius), _ptr(out),
                                         _ptr(out_off), _ptr(ws), wsb, _stream()), "geomae_points_pipeline")
        offs = out_off.cpu().tolist()          # sizes of the ragged outputs (runs in the loader, off the training stream)
        return [out[offs[b]:offs[b + 1]] for b in range(len(frames))]
