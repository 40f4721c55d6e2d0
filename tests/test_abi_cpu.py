"""CPU checks of the C-ABI boundary: the shared library loads and exports exactly the symbols that
include/geomae_hip.h declares (no compute calls -- there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "geomae_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(geomae_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    from geomae_amd.csrc import build as b
    so = b.build()
    assert os.path.exists(so)
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = set(re.findall(r"\bT (geomae_[a-z0-9_]+)", out))
    hdr = _header_symbols()
    assert hdr, "no symbols parsed from the header"
    assert set(hdr) <= exported, sorted(set(hdr) - exported)
    assert exported <= set(hdr), f"exported but undeclared: {sorted(exported - set(hdr))}"


def test_ctypes_table_matches_header_and_loads():
    from geomae_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    lib = _lib.load()
    assert lib.geomae_abi_version() == 1
    assert lib.geomae_last_error() is not None


def test_host_side_argument_checks_need_no_gpu():
    import ctypes
    from geomae_amd import _lib
    lib = _lib.load()
    out = (ctypes.c_int32 * 3)()
    assert lib.geomae_grid_size(_lib.f3((0.256, 0.256, 8)), _lib.f3((-51.2, -51.2, -5, 51.2, 51.2, 3)), out) == 0
    assert list(out) == [400, 400, 1]
    assert lib.geomae_grid_size(_lib.f3((0.064, 0.064, 1)), _lib.f3((-51.2, -51.2, -5, 51.2, 51.2, 3)), out) == 0
    assert list(out) == [1600, 1600, 8]
    assert lib.geomae_grid_size(_lib.f3((0.5, 0.5, 8)), _lib.f3((-51.2, -51.2, -5, 51.2, 51.2, 3)), out) == 0
    assert list(out) == [205, 205, 1]
    rc = lib.geomae_grid_size(_lib.f3((0.0, 0.5, 8)), _lib.f3((-51.2, -51.2, -5, 51.2, 51.2, 3)), out)
    assert rc < 0 and b"voxel_size" in lib.geomae_last_error()
    # bad arguments are rejected before any launch
    rc = lib.geomae_dynamic_voxelize(None, 10, 5, _lib.f3((1, 1, 1)), _lib.f3((0, 0, 0, 1, 1, 1)), None, None)
    assert rc < 0 and b"null" in lib.geomae_last_error()
    rc = lib.geomae_dynamic_voxelize(None, 0, 5, _lib.f3((1, 1, 1)), _lib.f3((0, 0, 0, 1, 1, 1)), None, None)
    assert rc == 0                                            # empty input is a no-op


def test_pretrain_engine_argument_checks_need_no_gpu():
    """The step-level entry points validate their host-side arguments before touching the device."""
    import ctypes
    from geomae_amd import _lib
    lib = _lib.load()
    c = _lib.GeomaePretrainConfig()
    assert lib.geomae_pretrain_workspace_bytes(ctypes.byref(c), 1000, 100) == -1          # all-zero config
    c.batch_size, c.num_features, c.num_heads, c.encoder_layers, c.decoder_layers = 4, 5, 8, 12, 4
    c.targets.grid_size[:] = [1, 400, 400]
    c.targets.ratio_low[:], c.targets.ratio_med[:] = [8, 4, 4], [4, 2, 2]
    c.window.window_shape[:], c.window.shift[:], c.window.bev_shape[:] = [12, 12], [6, 6], [400, 400]
    c.keep_fraction = 0.3
    small = lib.geomae_pretrain_workspace_bytes(ctypes.byref(c), 110_000, 30_000)
    big = lib.geomae_pretrain_workspace_bytes(ctypes.byref(c), 1_100_000, 120_000)
    assert 0 < small < big < 64 * 2 ** 30                      # ~2 GB for a single-sweep batch, well inside 288 GB
    assert lib.geomae_pretrain_workspace_bytes(ctypes.byref(c), 0, 10) == -1
    c.keep_fraction = 1.5
    assert lib.geomae_pretrain_workspace_bytes(ctypes.byref(c), 1000, 100) == -1 and b"keep_fraction" in lib.geomae_last_error()
    c.keep_fraction = 0.3
    c.targets.grid_size[:] = [2, 400, 400]
    assert lib.geomae_pretrain_workspace_bytes(ctypes.byref(c), 1000, 100) == -1 and b"top grid" in lib.geomae_last_error()
    assert not lib.geomae_pretrain_create(ctypes.byref(c), None, None, 0, 1000, 100, None)
    assert lib.geomae_pretrain_step(None, None, None, 1e-5, 1.0, 1, None) < 0 and b"null engine" in lib.geomae_last_error()
    assert lib.geomae_pretrain_result_offset(None, 0) == -1
    lib.geomae_pretrain_destroy(None)                          # a no-op, like free(NULL)


def test_missing_library_fails_loudly(tmp_path):
    from geomae_amd import _lib
    with pytest.raises(_lib.GeomaeLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_ops_refuse_cpu_tensors():
    import torch
    from geomae_amd import ops
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.dynamic_voxelize(torch.zeros(4, 5), torch.zeros(4, 3, dtype=torch.int32), (1, 1, 1), (0, 0, 0, 1, 1, 1))
