"""Re-generate ALL ELEVEN fixtures from /root/reference into a scratch directory and compare them, array by array, with
the committed tests/golden/*.npz (build container only; TEST INFRASTRUCTURE ONLY).  Nothing under tests/golden/ is written.
    PYTHONDONTWRITEBYTECODE=1 python oracle/verify_golden.py [round1 syncbn pipeline winops head fullsize c1 | all]
`round1` = oracle/make_golden.py's five files (g1_voxelize, g_pipeline_tiny, g_pipeline_full, g_hard_voxelize, g_finetune).
Exit code 0 = every array identical bit for bit, except the float arrays of the multi-threaded torch-CPU runs (syncbn,
fullsize, c1, round1's pipeline / finetune gradients), whose summation order varies with the thread schedule: those must
agree within 1e-6 of the array's largest magnitude (measured 1e-7)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GEN = dict(syncbn=("make_golden_syncbn.py", "g_syncbn_w2.npz"), pipeline=("make_golden_pipeline.py", "g_input_pipeline.npz"),
           winops=("make_golden_winops.py", "g_winops.npz"), head=("make_golden_head.py", "g_head.npz"),
           fullsize=("make_golden_fullsize.py", "g_fullsize.npz"), c1=("make_golden_fullsize.py c1", "g_pipeline_c1.npz"),
           round1=("make_golden.py", "g1_voxelize.npz g_pipeline_tiny.npz g_pipeline_full.npz g_hard_voxelize.npz g_finetune.npz"))
FLOAT_TOL = ("fullsize", "syncbn", "c1", "round1")


def main():
    which = [a for a in sys.argv[1:] if a in GEN] or ["syncbn", "pipeline", "winops", "head"]
    if "all" in sys.argv[1:]:
        which = list(GEN)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name in which:
            script, npzs = GEN[name]
            env = dict(os.environ, GEOMAE_GOLDEN_OUT=tmp, PYTHONDONTWRITEBYTECODE="1")
            script, *extra = script.split()
            subprocess.run([sys.executable, os.path.join(HERE, script)] + extra, env=env, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            for npz in npzs.split():
                new, old = np.load(os.path.join(tmp, npz)), np.load(os.path.join(ROOT, "tests", "golden", npz))
                assert sorted(new.files) == sorted(old.files), (name, npz, set(new.files) ^ set(old.files))
                diff = []
                for k in old.files:
                    a, b = old[k], new[k]
                    same = a.shape == b.shape and (np.array_equal(a, b) or (
                        name in FLOAT_TOL and a.dtype.kind == "f" and
                        float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) <= 1e-6 * max(float(np.abs(a).max()), 1e-30)))
                    if not same:
                        diff.append(k)
                print(f"{name} / {npz}: {len(old.files)} arrays, {len(diff)} differ {diff[:5]}")
                bad += len(diff)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
