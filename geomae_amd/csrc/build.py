"""Build libgeomae_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libgeomae_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-result"]
# Floating-point contraction: OFF where results are compared bit-for-bit with the oracle / torch (voxel indices,
# geometric targets, the input pipeline, the AdamW update); ON (fused multiply-add) in the MFMA kernels, whose
# parity is a tolerance anyway: 9-12 % fewer VALU instructions in the layer kernels.  sst_fused.hip / sst_ws.hip (the one-launch
# layer kernels) joined in round 6: same-box A/B of the step 1.673 / 1.673 / 1.674 ms with contraction off against 1.653 / 1.655 /
# 1.659 ms with it on (tools/ab_lib.py), the fused and engine-vs-reference tests green at unchanged bounds.
CONTRACT_FAST = {"sst_layer.hip", "window.hip", "heads_loss.hip", "vfe.hip", "sst_ws.hip", "sst_fused.hip"}


def flags_for(src):
    return FLAGS + ["-ffp-contract=" + ("fast" if os.path.basename(src) in CONTRACT_FAST else "off")]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(ROOT, "include", "geomae_hip.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src, os.path.abspath(__file__)] + hdrs):
            jobs.append([HIPCC] + flags_for(src) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or _newer(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
