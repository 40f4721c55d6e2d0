"""ctypes loader for the plain-C oracle (oracle/voxelize.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(HERE, "libgeomae_oracle.so")
    src = os.path.join(HERE, "voxelize.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "libgeomae_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def dynamic_voxelize_c(points, voxel_size, pc_range):
    p = np.ascontiguousarray(points, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    rg = np.asarray(pc_range, np.float32)
    out = np.empty((p.shape[0], 3), np.int32)
    f = lib().geomae_oracle_dynamic_voxelize
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    f(p.ctypes.data, p.shape[0], p.shape[1], vs.ctypes.data, rg.ctypes.data, out.ctypes.data)
    return out
