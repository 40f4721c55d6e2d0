"""Import the reference's Python hot path under stub modules (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE ONLY -- used by oracle/make_golden.py to generate the committed
fixtures under tests/golden/.  Nothing here runs on the GPU box (/root/reference does
not exist there) and nothing from the reference is copied: its files are loaded by path.

Un-vendored dependencies of the reference that cannot be imported here are replaced by
small stand-ins, each restating the dependency's published behaviour:
  * torch_scatter 2.x  scatter(reduce=mean|sum) / scatter_max  (call sites
    mmdet3d/ops/sst/sst_ops.py:30,32)                            -> _TorchScatterStub
  * spconv-cu113 2.1.21 get_indice_pairs_implicit_gemm(subm=True, ksize=[1,3,3])
    (call site detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py:192-207)
                                                                 -> _indice_pairs_subm_3x3
  * mmdet 2.20.0 CrossEntropyLoss(use_sigmoid=True) (call sites ssl.py:112-115,894-895)
                                                                 -> _SigmoidCE
  * mmcv decorators auto_fp16 / force_fp32 (inert: no fp16 key in the config)
Parity for those three stand-ins is therefore "unpinned" by the reference itself.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
REF = os.environ.get("GEOMAE_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, relpath, package=None):
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    if package:
        m.__package__ = package
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


# ---- torch_scatter stand-in -------------------------------------------------------
def _scatter(src, index, dim=0, reduce="sum", dim_size=None):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    out.index_add_(0, index, src)
    if reduce == "mean":
        cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
        out = out / cnt.view(-1, *([1] * (src.dim() - 1)))
    return out


def _scatter_max(src, index, dim=0, dim_size=None):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    # argmax (first point attaining the max), as torch_scatter returns
    is_max = src == out[index]
    pos = torch.arange(src.shape[0]).view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    cand = torch.where(is_max, pos, torch.full_like(pos, src.shape[0]))
    arg = torch.full(out.shape, src.shape[0], dtype=torch.long)
    arg = arg.scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
    return out, arg


# ---- spconv stand-in --------------------------------------------------------------
def _indice_pairs_subm_3x3(indices, batch_size, spatial_shape, algo=None, ksize=None,
                           stride=None, padding=None, dilation=None, out_padding=None,
                           subm=True, transpose=False, is_train=False, **kw):
    """pair[k, i] = row of the pillar at BEV offset k=(dy+1)*3+(dx+1) from pillar i, else -1."""
    assert subm and list(ksize) == [1, 3, 3]
    _, ny, nx = spatial_shape
    idx = indices.long()
    V = idx.shape[0]
    table = torch.full((batch_size * ny * nx,), -1, dtype=torch.long)
    table[idx[:, 0] * ny * nx + idx[:, 2] * nx + idx[:, 3]] = torch.arange(V)
    pair = torch.full((9, V), -1, dtype=torch.int32)
    k = 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            y = idx[:, 2] + dy
            x = idx[:, 3] + dx
            ok = (y >= 0) & (y < ny) & (x >= 0) & (x < nx)
            lin = idx[:, 0] * ny * nx + y.clamp(0, ny - 1) * nx + x.clamp(0, nx - 1)
            nb = torch.where(ok, table[lin], torch.full_like(lin, -1))
            pair[k] = nb.to(torch.int32)
            k += 1
    return (None, None, pair, None, None, None, None, None, None)


# ---- mmdet CrossEntropyLoss(use_sigmoid=True) stand-in ------------------------------
class _SigmoidCE(nn.Module):
    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, label):
        onehot = F.one_hot(label, pred.shape[-1]).to(pred.dtype)
        return self.loss_weight * F.binary_cross_entropy_with_logits(pred, onehot, reduction="mean")


def _build_loss(cfg):
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t == "CrossEntropyLoss":
        assert cfg.get("use_sigmoid")
        return _SigmoidCE(cfg.get("loss_weight", 1.0))
    if t == "SmoothL1Loss":
        return nn.SmoothL1Loss(reduction=cfg.get("reduction", "mean"))
    raise KeyError(t)


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _identity_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


_LOADED = {}


def load_reference():
    """Returns a namespace with the reference's hot-path modules and the real CPU voxelizer."""
    if _LOADED:
        return types.SimpleNamespace(**_LOADED)
    _mod("ipdb", set_trace=lambda *a, **k: None)
    _mod("torch_scatter", scatter=_scatter, scatter_max=_scatter_max)
    _mod("mmcv")
    def _build_norm_layer(cfg, num_features):
        # mmcv.cnn.build_norm_layer restated for the one norm type on the path:
        # 'naiveSyncBN1d' == nn.BatchNorm1d when world_size == 1 (ops/norm.py:58-59)
        if cfg["type"] in ("naiveSyncBN2d", "BN", "BN2d"):      # == nn.BatchNorm2d when world_size == 1 (ops/norm.py:121-122)
            return "bn", nn.BatchNorm2d(num_features, eps=cfg.get("eps", 1e-5), momentum=cfg.get("momentum", 0.1))
        assert cfg["type"] in ("naiveSyncBN1d", "BN1d")
        return "bn", nn.BatchNorm1d(num_features, eps=cfg.get("eps", 1e-5),
                                    momentum=cfg.get("momentum", 0.1))

    def _build_conv_layer(cfg, *args, **kwargs):
        # mmcv.cnn.build_conv_layer restated for cfg type 'Conv2d': the remaining cfg keys are constructor kwargs
        c = dict(cfg or dict(type="Conv2d"))
        assert c.pop("type") == "Conv2d"
        return nn.Conv2d(*args, **kwargs, **c)

    _mod("mmcv.cnn", build_conv_layer=_build_conv_layer, build_norm_layer=_build_norm_layer,
         NORM_LAYERS=_Registry())
    _mod("mmcv.runner", auto_fp16=_identity_decorator, force_fp32=_identity_decorator)
    _mod("mmdet")
    _mod("mmdet.models", BACKBONES=_Registry(), DETECTORS=_Registry())
    spconv = _mod("spconv")
    _mod("spconv.pytorch")
    _mod("spconv.pytorch.ops", get_indice_pairs=None,
         get_indice_pairs_implicit_gemm=_indice_pairs_subm_3x3)
    _mod("spconv.core", ConvAlgo=types.SimpleNamespace(MaskImplicitGemm=0))
    m3 = _mod("mmdet3d")
    m3.__path__ = []
    ops = _mod("mmdet3d.ops", spconv=spconv, Voxelization=None, Voxelization_with_flag=None,
               points_in_boxes_cpu=None, points_in_boxes_gpu=None)
    ops.__path__ = []
    sst_ops = _load("mmdet3d.ops.sst.sst_ops", "mmdet3d/ops/sst/sst_ops.py")
    ops.DynamicScatter = lambda *a, **k: None      # constructed, never called by DynamicScatterVFE
    ops.make_sparse_convmodule = None
    for n in ("flat2window", "window2flat", "scatter_v2", "get_inner_win_inds",
              "make_continuous_inds", "get_flat2win_inds"):
        setattr(ops, n, getattr(sst_ops, n))
    models = _mod("mmdet3d.models")
    models.__path__ = []
    _mod("mmdet3d.models.sst").__path__ = []
    _mod("mmdet3d.models.builder", build_loss=_build_loss, build_voxel_encoder=lambda cfg: None,
         VOXEL_ENCODERS=_Registry(), MIDDLE_ENCODERS=_Registry(), build_fusion_layer=None)
    models.builder = sys.modules["mmdet3d.models.builder"]
    _mod("mmdet3d.models.detectors").__path__ = []
    _mod("mmdet3d.models.detectors.voxelnet", VoxelNet=object)

    class SingleStage3DDetector(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _mod("mmdet3d.models.detectors.single_stage", SingleStage3DDetector=SingleStage3DDetector)
    _mod("mmdet3d.core", bbox3d2result=None, merge_aug_bboxes_3d=None)
    blk = _load("mmdet3d.models.sst.sst_basic_block", "mmdet3d/models/sst/sst_basic_block.py")
    bb = _load("mmdet3d.models.backbones.multi_mae_sst_spearate_top_only",
               "mmdet3d/models/backbones/multi_mae_sst_spearate_top_only.py")
    ssl = _load("mmdet3d.models.detectors.multi_sub_voxel_dynamic_voxelnet_ssl",
                "mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py",
                package="mmdet3d.models.detectors")
    _mod("mmdet3d.models.voxel_encoders").__path__ = []
    _load("mmdet3d.models.voxel_encoders.utils", "mmdet3d/models/voxel_encoders/utils.py",
          package="mmdet3d.models.voxel_encoders")
    vfe = _load("mmdet3d.models.voxel_encoders.voxel_encoder",
                "mmdet3d/models/voxel_encoders/voxel_encoder.py",
                package="mmdet3d.models.voxel_encoders")
    # fine-tune path (N1): SSTInputLayer + SSTSecondPretrainedv1
    _mod("mmdet3d.models.middle_encoders").__path__ = []
    mid = _load("mmdet3d.models.middle_encoders.sst_input_layer", "mmdet3d/models/middle_encoders/sst_input_layer.py",
                package="mmdet3d.models.middle_encoders")
    _mod("mmdet3d.models.backbones").__path__ = []
    ft = _load("mmdet3d.models.backbones.sst_second_pretrained_v1", "mmdet3d/models/backbones/sst_second_pretrained_v1.py",
               package="mmdet3d.models.backbones")
    _LOADED.update(mid=mid, ft=ft)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import build_ref
    build_ref.build()
    voxel_layer = build_ref.load()
    _LOADED.update(sst_ops=sst_ops, blk=blk, bb=bb, ssl=ssl, vfe=vfe, voxel_layer=voxel_layer)
    return types.SimpleNamespace(**_LOADED)


def make_detector(ref, cfg_model):
    """Instance of the reference detector with only the attributes the target methods read
    (SURVEY Appendix B step 5); the VFE is restated separately (voxel_encoder.py needs mmcv)."""
    cls = ref.ssl.MultiSubVoxelDynamicVoxelNetSSL
    det = cls.__new__(cls)
    nn.Module.__init__(det)
    m = cfg_model
    det.grid_size = m["grid_size"]
    det.sub_voxel_ratio_low = m["sub_voxel_ratio_low"]
    det.sub_voxel_ratio_med = m["sub_voxel_ratio_med"]
    det.voxel_size = m["voxel_layer"]["voxel_size"]
    det.sub_voxel_size_low = m["sub_voxel_layer_low"]["voxel_size"]
    det.sub_voxel_size_med = m["sub_voxel_layer_med"]["voxel_size"]
    det.point_cloud_range = m["voxel_layer"]["point_cloud_range"]
    det.random_mask_ratio = m["random_mask_ratio"]
    det.norm_curv = True
    det.spatial_shape = m["spatial_shape"]
    det.mse_loss = m["mse_loss"]
    det.nor_usr_sml1 = None
    det.cls_sub_voxel = m["cls_sub_voxel"]
    for k in ("loss_ratio_low", "loss_ratio_med", "loss_ratio_top", "loss_ratio_low_nor",
              "cls_loss_ratio_low", "cls_loss_ratio_med"):
        setattr(det, k, m[k])
    det.cls_loss = _SigmoidCE(1.0)
    det.normalize_sub_voxel = m["normalize_sub_voxel"]
    det.use_focal_mask = None
    return det
