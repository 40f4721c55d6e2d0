"""Minimal Registry / build_from_cfg so that configs/mae_sst/*.py resolve every `type=` string.

mmcv / mmdet are not installed here (and must be assumed absent on the GPU box); this restates
the small part of mmcv.utils.Registry the reference relies on (mmdet3d/models/builder.py:5-98:
DETECTORS / BACKBONES come from mmdet.models, VOXEL_ENCODERS = MODELS, norm layers register into
mmcv.cnn.NORM_LAYERS, mmdet3d/ops/norm.py:28).
"""
import inspect


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, module=None, force=False):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def _register(self, cls, name, force):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if n in self._modules and not force:
                raise KeyError(f"{n} is already registered in {self.name}")
            self._modules[n] = cls

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args or None)

    def __repr__(self):
        return f"Registry({self.name}, {sorted(self._modules)})"


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise KeyError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop("type")
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f"{t} is not in the {registry.name} registry")
    if not (inspect.isclass(cls) or callable(cls)):
        raise TypeError(f"type must be a str or class, got {type(cls)}")
    return cls(**args)


MODELS = Registry("models")
DETECTORS = MODELS
BACKBONES = MODELS
VOXEL_ENCODERS = MODELS
MIDDLE_ENCODERS = MODELS
NECKS = MODELS
HEADS = MODELS
LOSSES = Registry("losses")
NORM_LAYERS = Registry("norm layer")
DATASETS = Registry("dataset")
PIPELINES = Registry("pipeline")


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d/models/builder.py:47-58."""
    return DETECTORS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d/models/builder.py:75-84 (segmentors are out of scope)."""
    return build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_voxel_encoder(cfg):
    return VOXEL_ENCODERS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_norm_layer(cfg, num_features):
    """mmcv.cnn.build_norm_layer restated: returns (name, layer)."""
    args = dict(cfg)
    t = args.pop("type")
    cls = NORM_LAYERS.get(t)
    if cls is None:
        raise KeyError(f"{t} is not in the norm layer registry")
    args.pop("requires_grad", None)
    return t, cls(num_features, **args)
