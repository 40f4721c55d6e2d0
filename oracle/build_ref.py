"""Build recipe for oracle/_ref: the REFERENCE's own CPU voxelizer, compiled in place.

TEST INFRASTRUCTURE ONLY.  Compiles the reference's pybind module `voxel_layer`
(/root/reference/mmdet3d/ops/voxel/src/{voxelization.cpp,voxelization_cpu.cpp,
scatter_points_cpu.cpp}; bound functions voxelization.cpp:5-11) from the sources where
they lie, with g++ against the installed torch headers (no WITH_CUDA, so only the CPU
paths exist: dynamic_voxelize, hard_voxelize).  Output goes to oracle/_ref/ only, which
is git-ignored (kept out of history) but NOT listed in .gpurunignore: the built binary is
part of a `gpurun` snapshot like the in-tree HIP library; reference SOURCES never leave
this container.

Built only where /root/reference exists (the build container), where it validates the
oracle's plain-C voxelizer and generates the committed fixtures (tests/golden/*.npz).
Nothing that runs on the GPU box -- `pytest -m gpu`, `__graft_entry__.smoke()`, `bench.py`
-- loads it: those compare against the fixtures and the oracle (its only consumer is the
CPU test tests/test_oracle_golden.py, which skips when the binary is absent).
"""
import os
import subprocess
import sys
import sysconfig

REF = os.environ.get("GEOMAE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "voxel_layer_ref"


def out_path():
    return os.path.join(OUT_DIR, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False, verbose=False):
    src_dir = os.path.join(REF, "mmdet3d", "ops", "voxel", "src")
    srcs = [os.path.join(src_dir, f) for f in
            ("voxelization.cpp", "voxelization_cpu.cpp", "scatter_points_cpu.cpp")]
    if not all(os.path.exists(s) for s in srcs):
        return None  # reference not mounted here (GPU box): use the prebuilt file
    out = out_path()
    if os.path.exists(out) and not force:
        if all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
            return out
    os.makedirs(OUT_DIR, exist_ok=True)
    from torch.utils import cpp_extension as ce
    import torch
    inc = ce.include_paths()
    lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-w", "-shared", "-fPIC", "-std=c++17",
           f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-I" + p for p in inc]
    cmd += srcs
    cmd += ["-L" + lib_dir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
            "-Wl,-rpath," + lib_dir, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    return out


def load():
    """Import the prebuilt reference module (or None when absent)."""
    out = out_path()
    if not os.path.exists(out):
        return None
    import importlib.util
    import torch  # noqa: F401  (must be imported before the extension)
    spec = importlib.util.spec_from_file_location(NAME, out)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built:" if p else "reference not present; nothing built", p)
