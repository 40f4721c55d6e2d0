"""Seeds / generator arguments of the full-size fixtures (tests/golden/g_fullsize.npz): shared by the generator
(oracle/make_golden_fullsize.py, build container) and the GPU tests; data only, nothing from the reference."""
NUS = dict(range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], top=(0.256, 0.256, 8), med=(0.128, 0.128, 2), low=(0.064, 0.064, 1),
           grid=(1, 400, 400))
WAYMO = dict(range=[-74.88, -74.88, -2.0, 74.88, 74.88, 4.0], top=(0.32, 0.32, 6), med=(0.16, 0.16, 1.5),
             low=(0.08, 0.08, 0.75), grid=(1, 468, 468))
WAYMO_FRAME = dict(beams=64, n_az=3100, pc_range=(-74.88, -74.88, -2.0, 74.88, 74.88, 4.0), elev=(-17.6, 2.4), n_cyl=90,
                   max_range=110.0)
CASES = dict(
    c2=(NUS, [dict(seed=2000 + i) for i in range(4)]),          # BASELINE config 2: 4 single-sweep frames
    c3=(NUS, [dict(seed=3000, sweeps=10)]),                     # config 3: one 10-sweep frame
    c4=(WAYMO, [dict(seed=4000, **WAYMO_FRAME)]),               # config 4: Waymo geometry, ~180 k points
)

# BASELINE config 1: 0.5 m pillars -> fp32 grid ceil(204.8) = 205 (NOT a multiple of the 12-pillar window),
# sub-voxels 0.25 x 0.25 x 2 and 0.125 x 0.125 x 1, SST-tiny = 1 + 1 + 1 blocks; the 16 k uniform cloud, a LiDAR-ring
# cloud, and both as one batch of two (tests/golden/g_pipeline_c1.npz).  `gen` names the geomae_amd.synth generator.
C1 = dict(range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], top=(0.5, 0.5, 8), med=(0.25, 0.25, 2), low=(0.125, 0.125, 1),
          grid=(1, 205, 205), blocks=(1, 1))
C1_UNIFORM, C1_LIDAR = dict(gen="uniform_cloud", seed=0, n=16000), dict(seed=1, beams=16, n_az=800)
CASES_C1 = dict(c1u=(C1, [C1_UNIFORM]), c1l=(C1, [C1_LIDAR]), c1b=(C1, [C1_UNIFORM, C1_LIDAR]))


def make_frames(specs):
    from geomae_amd import synth
    out = []
    for kw in specs:
        kw = dict(kw)
        out.append(getattr(synth, kw.pop("gen", "lidar_frame"))(**kw))
    return out
