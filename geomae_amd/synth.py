"""Seeded synthetic LiDAR sweeps shaped like the nuScenes input contract.

There is no dataset on the GPU box, so bench/tests use these clouds.
Layout per point: [x, y, z, intensity, dt] fp32 -- the contract produced by the reference's
LoadPointsFromFile / LoadPointsFromMultiSweeps (mmdet3d/datasets/pipelines/loading.py:100-237:
remove_close |x|<1 & |y|<1, key-frame dt = 0) followed by PointsRangeFilter (strict open
interval, mmdet3d/core/points/base_points.py:223-228) and PointShuffle.

Host-side numpy only; nothing here is on the timed path.
"""
import numpy as np

NUS_RANGE = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)


def _scene(rng, n_cyl, extent):
    cx = rng.uniform(-extent, extent, n_cyl)
    cy = rng.uniform(-extent, extent, n_cyl)
    keep = np.hypot(cx, cy) > 3.0
    cx, cy = cx[keep], cy[keep]
    r = rng.uniform(0.3, 2.5, cx.size)
    h = rng.uniform(0.8, 4.0, cx.size)
    return cx, cy, r, h


def _cast(rng, origin_xy, yaw, scene, beams, n_az, sensor_h, max_range, noise):
    cx, cy, r, h = scene
    elev = np.deg2rad(beams)[:, None]
    az = (np.arange(n_az) * (2 * np.pi / n_az) + yaw)[None, :]
    dx = (np.cos(elev) * np.cos(az)).ravel()
    dy = (np.cos(elev) * np.sin(az)).ravel()
    dz = (np.sin(elev) * np.ones_like(az)).ravel()
    t_best = np.full(dx.shape, np.inf)
    # ground plane z = -sensor_h
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dz < 0, -sensor_h / dz, np.inf)
    t_best = np.minimum(t_best, tg)
    refl = np.full(dx.shape, 20.0)
    # vertical cylinders (ray/circle in the xy plane)
    ox, oy = origin_xy
    dxy2 = dx * dx + dy * dy
    for k in range(cx.size):
        fx, fy = ox - cx[k], oy - cy[k]
        b = fx * dx + fy * dy
        c = fx * fx + fy * fy - r[k] * r[k]
        disc = b * b - dxy2 * c
        ok = disc > 0
        t = np.where(ok, (-b - np.sqrt(np.where(ok, disc, 0.0))) / np.maximum(dxy2, 1e-12), np.inf)
        z = t * dz
        ok = ok & (t > 0.5) & (z > -sensor_h) & (z < -sensor_h + h[k])
        t = np.where(ok, t, np.inf)
        hit = t < t_best
        t_best = np.where(hit, t, t_best)
        refl = np.where(hit, 40.0 + 3.0 * (k % 60), refl)
    valid = np.isfinite(t_best) & (t_best < max_range)
    t = t_best[valid] + rng.normal(0.0, noise, valid.sum())
    pts = np.stack([ox + t * dx[valid], oy + t * dy[valid], t * dz[valid]], axis=1)
    inten = np.clip(refl[valid] + rng.normal(0, 5.0, valid.sum()), 0, 255)
    return pts, inten


def lidar_frame(seed, sweeps=1, beams=32, n_az=1085, pc_range=NUS_RANGE, sensor_h=1.84,
                elev=(-30.67, 10.67), n_cyl=60, noise=0.02, max_range=100.0, shuffle=True):
    """One training sample: `sweeps` sweeps merged into the key-frame's coordinates.

    Returns float32 [N, 5].  sweeps=1 ~ 27-30k points (BASELINE config 2), sweeps=10 ~ 267k
    (config 3).  beams=64, n_az=2650, pc_range=(+-74.88, [-2,4]) approximates config 4.
    """
    rng = np.random.default_rng(seed)
    extent = max(abs(pc_range[0]), abs(pc_range[3]))
    scene = _scene(rng, n_cyl, extent)
    beam_angles = np.linspace(elev[0], elev[1], beams)
    out = []
    for s in range(sweeps):
        ego = rng.normal(0, 0.4, 2) * (s > 0) + np.array([0.5 * s, 0.0]) * (s > 0)
        yaw = rng.uniform(0, 2 * np.pi)
        pts, inten = _cast(rng, ego, yaw, scene, beam_angles, n_az, sensor_h, max_range, noise)
        close = (np.abs(pts[:, 0]) < 1.0) & (np.abs(pts[:, 1]) < 1.0)
        pts, inten = pts[~close], inten[~close]
        dt = np.full((pts.shape[0], 1), 0.05 * s)
        out.append(np.concatenate([pts, inten[:, None], dt], axis=1))
    p = np.concatenate(out, axis=0).astype(np.float32)
    lo = np.asarray(pc_range[:3], np.float32)
    hi = np.asarray(pc_range[3:], np.float32)
    keep = np.all((p[:, :3] > lo) & (p[:, :3] < hi), axis=1)
    p = p[keep]
    if shuffle:
        p = p[rng.permutation(p.shape[0])]
    return np.ascontiguousarray(p)


def uniform_cloud(seed, n=16000, pc_range=NUS_RANGE):
    """BASELINE config 1: uniform points in the range box, intensity U(0,255), dt = 0."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(pc_range[:3])
    hi = np.asarray(pc_range[3:])
    xyz = rng.uniform(lo, hi, (n, 3))
    inten = rng.uniform(0, 255, (n, 1))
    return np.concatenate([xyz, inten, np.zeros((n, 1))], axis=1).astype(np.float32)


def boundary_cloud(pc_range=NUS_RANGE, voxel=(0.064, 0.064, 1.0)):
    """Edge cases for the bit-exact voxelizer: every face +-1 ulp, exact cell multiples,
    out-of-range points on both sides (this fork clamps them, SURVEY item 5)."""
    lo = np.asarray(pc_range[:3], np.float32)
    hi = np.asarray(pc_range[3:], np.float32)
    vs = np.asarray(voxel, np.float32)
    pts = []
    mid = ((lo + hi) / 2).astype(np.float32)
    for d in range(3):
        for base in (lo[d], hi[d]):
            for v in (np.nextafter(base, np.float32(-1e9)), base, np.nextafter(base, np.float32(1e9)),
                      base - np.float32(3.0), base + np.float32(3.0)):
                p = mid.copy()
                p[d] = v
                pts.append(p)
        # exact multiples of the finest cell and their fp32 neighbours
        for k in (0, 1, 2, 3, 7, 100, 799, 1599, 1600):
            e = np.float32(lo[d] + np.float32(k) * vs[d])
            for v in (np.nextafter(e, np.float32(-1e9)), e, np.nextafter(e, np.float32(1e9))):
                p = mid.copy()
                p[d] = v
                pts.append(p)
    pts = np.asarray(pts, np.float32)
    extra = np.zeros((pts.shape[0], 2), np.float32)
    return np.concatenate([pts, extra], axis=1)
