"""Generate tests/golden/g_winops.npz from the REFERENCE's mmdet3d/ops/sst/sst_ops.py (build container only):
make_continuous_inds (:371-388), get_inner_win_inds (:271-319), get_flat2win_inds (:57-96), flat2window (:98-135),
window2flat (:225-251) on seeded window ids with three drop levels.  TEST INFRASTRUCTURE ONLY; data only.
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_winops.py
The in-window order of the reference depends on an unstable sort, so the fixture stores order-free facts: continuous
ids, tokens per window, and for every padded [W, T, C] tensor the per-window SORTED row sums (+ the zero padding)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import ref_import  # noqa: E402

DROP = {0: dict(max_tokens=8, drop_range=(0, 8)), 1: dict(max_tokens=20, drop_range=(8, 20)),
        2: dict(max_tokens=48, drop_range=(20, 100000))}


def inputs(seed=21, n_windows=160, max_id=4000):
    """Window ids with sizes 1..48, scattered over [0, max_id); levels from the sizes (like drop_single_shift)."""
    rs = np.random.RandomState(seed)
    ids = np.sort(rs.choice(max_id, n_windows, replace=False))
    sizes = rs.randint(1, 49, n_windows)
    win = np.repeat(ids, sizes)
    perm = rs.permutation(win.shape[0])
    win = win[perm]
    cnt = np.bincount(win, minlength=max_id)[win]
    lvl = np.where(cnt <= 8, 0, np.where(cnt <= 20, 1, 2))
    feat = rs.standard_normal((win.shape[0], 16)).astype(np.float32)
    return win.astype(np.int64), lvl.astype(np.int64), feat


def main():
    ref = ref_import.load_reference()
    S = ref.sst_ops
    win, lvl, feat = inputs()
    win_t, lvl_t, feat_t = torch.as_tensor(win), torch.as_tensor(lvl), torch.as_tensor(feat)
    out = dict(n=np.int64(win.shape[0]))
    out["conti_all"] = S.make_continuous_inds(win_t).numpy()
    inner = S.get_inner_win_inds(win_t).numpy()
    out["inner_max_per_window"] = np.array([inner[win == w].max() for w in np.unique(win)], np.int64)
    inds = S.get_flat2win_inds(win_t, lvl_t, DROP, debug=True)
    f3d = S.flat2window(feat_t, lvl_t, inds, DROP)
    back = S.window2flat(f3d, inds)
    assert torch.equal(back, feat_t)
    for dl in DROP:
        assert dl in inds
        flat2win, where = inds[dl]
        T = DROP[dl]["max_tokens"]
        out[f"l{dl}.where"] = where[0].numpy()
        out[f"l{dl}.window_of_token"] = (flat2win // T).numpy()
        t = f3d[dl]
        out[f"l{dl}.shape"] = np.array(t.shape)
        out[f"l{dl}.sorted_rowsums"] = np.sort(t.double().sum(-1).numpy(), axis=1)
        out[f"l{dl}.nonzero_rows"] = (t.abs().sum(-1) > 0).sum(1).numpy()
    dst = os.path.join(os.environ.get("GEOMAE_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "g_winops.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
