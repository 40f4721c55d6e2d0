"""Python-file configs with `_base_` inheritance and `_delete_`, as mmcv.Config.fromfile loads them
(used by tools/train.py:101 of the reference).  Enough for configs/mae_sst/*.py and their three
_base_ files to load unchanged."""
import os
import runpy
import types


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(base, child):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = v
    return out


def _load(path):
    path = os.path.abspath(path)
    ns = runpy.run_path(path)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


class Config:
    def __init__(self, cfg_dict, filename=None):
        object.__setattr__(self, "_cfg", _wrap(cfg_dict))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load(filename), filename)

    def __getattr__(self, k):
        return getattr(object.__getattribute__(self, "_cfg"), k)

    def __getitem__(self, k):
        return self._cfg[k]

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def merge_from_dict(self, options):
        """--cfg-options dotted overrides (tools/train.py:62-71)."""
        for key, val in options.items():
            d = self._cfg
            parts = key.split(".")
            for p in parts[:-1]:
                d = d[int(p)] if isinstance(d, list) else d.setdefault(p, ConfigDict())
            if isinstance(d, list):
                d[int(parts[-1])] = val
            else:
                d[parts[-1]] = _wrap(val)
