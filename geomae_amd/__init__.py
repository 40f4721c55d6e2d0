"""geomae_amd -- MI355X-native GeoMAE-SST pre-training hot path.

Hand-written gfx950 kernels (libgeomae_hip.so, C ABI in include/geomae_hip.h) behind the
reference's operator / registry names.  Importing the package needs no GPU; any compute call
fails loudly if the HIP library is missing (there is no CPU fallback).
"""
from .registry import (BACKBONES, DETECTORS, LOSSES, MODELS, NORM_LAYERS, VOXEL_ENCODERS, Registry,  # noqa: F401
                       build_backbone, build_detector, build_from_cfg, build_loss, build_model,
                       build_norm_layer, build_voxel_encoder)
from .config import Config  # noqa: F401
from . import norm, losses  # noqa: F401  (register naiveSyncBN1d, CrossEntropyLoss, SmoothL1Loss)
from .voxel_encoder import DynamicScatterVFE  # noqa: F401
from .sst import BasicShiftBlock, EncoderLayer, MultiMAESSTSPChoose, WindowAttention  # noqa: F401
from .detector import MultiSubVoxelDynamicVoxelNetSSL  # noqa: F401
from .finetune import DynamicVoxelNet, SECONDFPN, SSTInputLayer, SSTSecondPretrainedv1  # noqa: F401
from .dense_head import Anchor3DHead  # noqa: F401

__version__ = "0.1.0"
