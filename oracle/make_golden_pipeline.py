"""Generate tests/golden/g_input_pipeline.npz by running the REFERENCE's own train pipeline classes (build container only).

TEST INFRASTRUCTURE ONLY.  Loaded by path from /root/reference under stub modules: LoadPointsFromFile,
LoadPointsFromMultiSweeps (datasets/pipelines/loading.py:184-233, _remove_close :162-182), GlobalRotScaleTrans,
RandomFlip3D, PointsRangeFilter, PointShuffle (datasets/pipelines/transforms_3d.py:77-160, 622-757, 770-797, 849-884)
and the real BasePoints / LiDARPoints (core/points/base_points.py, lidar_points.py).  The key frame and the sweeps go
through real .bin files (np.fromfile, the nuScenes on-disk layout: fp32 x 5 per point).
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_pipeline.py

Stand-ins for what the reference imports from un-vendored packages (structure only, no point math):
  * mmdet.datasets.pipelines.RandomFlip (mmdet 2.20): the base class of RandomFlip3D; its __call__ draws the 2-D flip
    decision with ONE np.random.choice before RandomFlip3D draws its own two numbers -- kept, so that the recorded
    np.random stream is the one a real run consumes ("parity unpinned" for that single draw);
  * box_type_3d: with no boxes, RandomFlip3D flips the points through an empty box object whose flip() hands them to
    points.flip(direction) (core/bbox/structures/lidar_box3d.py:191-200).
The fixture stores, per frame: the random decisions the reference made (recovered from its result dict and from a
recording wrapper around np.random), the filtered points BEFORE PointShuffle (concatenation order) and the shuffled
order's row checksum.  Inputs are regenerated from seeds by tests/test_pipeline_cpu.py::make_frames."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
REF = os.environ.get("GEOMAE_REFERENCE", "/root/reference")
RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, relpath, package=None):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    if package:
        m.__package__ = package
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


class _FileClient:
    def __init__(self, **kw):
        pass

    def get(self, path):
        raise ConnectionError           # -> the reference's np.fromfile branch (loading.py:393-399)


def _check_file_exist(path):
    assert os.path.isfile(path), path


class _RandomFlipBase:
    """Stand-in for mmdet 2.20 RandomFlip (un-vendored): stores flip_ratio, and __call__ makes the one 2-D draw."""

    def __init__(self, flip_ratio=None, direction="horizontal"):
        self.flip_ratio, self.direction = flip_ratio, direction

    def __call__(self, results):
        if "flip" not in results:
            cur = np.random.choice([self.direction, None], p=[self.flip_ratio, 1 - self.flip_ratio]) \
                if self.flip_ratio is not None else None
            results["flip"] = cur is not None
        results.setdefault("flip_direction", self.direction)
        return results


class _EmptyBoxes:
    def __init__(self, arr):
        assert len(arr) == 0

    def flip(self, bev_direction="horizontal", points=None):
        points.flip(bev_direction)      # core/bbox/structures/lidar_box3d.py:198-200
        return points


def load_reference_pipeline():
    _mod("mmcv", is_tuple_of=lambda seq, t: isinstance(seq, tuple) and all(isinstance(v, t) for v in seq),
         FileClient=_FileClient, check_file_exist=_check_file_exist)
    _mod("mmcv.utils", build_from_cfg=None)
    _mod("mmdet")
    _mod("mmdet.datasets")
    _mod("mmdet.datasets.builder", PIPELINES=_Registry())
    _mod("mmdet.datasets.pipelines", LoadAnnotations=object, LoadImageFromFile=object, RandomFlip=_RandomFlipBase)
    m3 = _mod("mmdet3d")
    m3.__path__ = []
    core = _mod("mmdet3d.core", VoxelGenerator=None)
    core.__path__ = []
    _mod("mmdet3d.core.bbox", box_np_ops=None)
    pts_pkg = _mod("mmdet3d.core.points")
    pts_pkg.__path__ = []
    base = _load("mmdet3d.core.points.base_points", "mmdet3d/core/points/base_points.py", "mmdet3d.core.points")
    lidar = _load("mmdet3d.core.points.lidar_points", "mmdet3d/core/points/lidar_points.py", "mmdet3d.core.points")
    pts_pkg.BasePoints, pts_pkg.LiDARPoints = base.BasePoints, lidar.LiDARPoints
    pts_pkg.get_points_type = lambda t: {"LIDAR": lidar.LiDARPoints}[t]
    ds = _mod("mmdet3d.datasets")
    ds.__path__ = []
    _mod("mmdet3d.datasets.builder", OBJECTSAMPLERS=_Registry())
    pl = _mod("mmdet3d.datasets.pipelines")
    pl.__path__ = []
    _mod("mmdet3d.datasets.pipelines.data_augment_utils", noise_per_object_v3_=None)
    loading = _load("mmdet3d.datasets.pipelines.loading", "mmdet3d/datasets/pipelines/loading.py", "mmdet3d.datasets.pipelines")
    tf = _load("mmdet3d.datasets.pipelines.transforms_3d", "mmdet3d/datasets/pipelines/transforms_3d.py",
               "mmdet3d.datasets.pipelines")
    return loading, tf


class _Recorder:
    """np.random with every call of the functions the pipeline uses logged (name, result)."""

    def __init__(self):
        self.log = []
        self._orig = {k: getattr(np.random, k) for k in ("choice", "uniform", "normal", "rand")}

    def __enter__(self):
        for k, f in self._orig.items():
            setattr(np.random, k, (lambda name, fn: lambda *a, **kw: self._rec(name, fn(*a, **kw)))(k, f))
        return self

    def _rec(self, name, val):
        self.log.append((name, val))
        return val

    def __exit__(self, *a):
        for k, f in self._orig.items():
            setattr(np.random, k, f)


def run_reference(frame, tmpdir, tag, sweeps_num, seed):
    loading, tf = load_reference_pipeline()
    key_path = os.path.join(tmpdir, f"{tag}_key.bin")
    np.asarray(frame["points"], np.float32).tofile(key_path)
    sweeps = []
    for k, sw in enumerate(frame["sweeps"]):
        p = os.path.join(tmpdir, f"{tag}_sw{k}.bin")
        np.asarray(sw["points"], np.float32).tofile(p)
        sweeps.append(dict(data_path=p, timestamp=sw["timestamp"], sensor2lidar_rotation=sw["sensor2lidar_rotation"],
                           sensor2lidar_translation=sw["sensor2lidar_translation"]))
    results = dict(pts_filename=key_path, sweeps=sweeps, timestamp=frame["timestamp"], bbox3d_fields=[],
                   box_type_3d=_EmptyBoxes, img_fields=[], pts_mask_fields=[], pts_seg_fields=[])
    # the train_pipeline of configs/mae_sst/...6x_1e-5.py:167-197
    stages = [loading.LoadPointsFromFile(coord_type="LIDAR", load_dim=5, use_dim=5),
              loading.LoadPointsFromMultiSweeps(sweeps_num=sweeps_num, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True,
                                                remove_close=True),
              tf.GlobalRotScaleTrans(rot_range=[-0.3925, 0.3925], scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0]),
              tf.RandomFlip3D(sync_2d=False, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
              tf.PointsRangeFilter(point_cloud_range=RANGE)]
    np.random.seed(seed)
    torch.manual_seed(seed)
    with _Recorder() as rec:
        for st in stages:
            results = st(results)
    before_shuffle = results["points"].tensor.clone().numpy()
    results = tf.PointShuffle()(results)
    shuffled = results["points"].tensor.numpy()
    return results, rec.log, before_shuffle, shuffled


def main():
    from test_pipeline_cpu import make_frames
    out = {}
    cases = [("a", 1, (3, 0), 2, 4000, 11), ("b", 2, (12, 1), 9, 1500, 12)]      # (tag, frame seed, sweeps per frame, sweeps_num, pts, rng seed)
    with tempfile.TemporaryDirectory() as tmp:
        for tag, fseed, n_sweeps, sweeps_num, n_pts, seed in cases:
            frames = make_frames(fseed, n_frames=2, n_sweeps=n_sweeps, n_pts=n_pts)
            out[f"{tag}.case"] = np.array([fseed, n_sweeps[0], n_sweeps[1], sweeps_num, n_pts, seed], np.int64)
            for b, fr in enumerate(frames):
                res, log, pts, shuffled = run_reference(fr, tmp, f"{tag}{b}", sweeps_num, seed + 100 * b)
                names = [n for n, _ in log]
                choices = np.asarray(log[0][1], np.int64) if (names and names[0] == "choice" and np.ndim(log[0][1]) == 1) \
                    else np.arange(min(len(fr["sweeps"]), sweeps_num), dtype=np.int64)
                uni = [v for n, v in log if n == "uniform"]
                k = f"{tag}.f{b}."
                out[k + "rng_calls"] = np.array(names)
                out[k + "sweep_choices"] = choices
                out[k + "rotation"] = np.float64(uni[0])
                out[k + "scale"] = np.float64(res["pcd_scale_factor"])
                assert np.float64(uni[1]) == out[k + "scale"]
                out[k + "translation"] = np.asarray(res["pcd_trans"], np.float64)
                out[k + "flip_horizontal"] = np.bool_(res["pcd_horizontal_flip"])
                out[k + "flip_vertical"] = np.bool_(res["pcd_vertical_flip"])
                out[k + "points"] = pts
                assert shuffled.shape == pts.shape
                out[k + "shuffled_rowsum"] = np.sort(shuffled.astype(np.float64).sum(1))
                print(tag, b, "sweeps", len(fr["sweeps"]), "->", pts.shape, "rng", names, "rot", float(uni[0]),
                      "flip", bool(res["pcd_horizontal_flip"]), bool(res["pcd_vertical_flip"]))
    dst = os.path.join(os.environ.get("GEOMAE_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "g_input_pipeline.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
