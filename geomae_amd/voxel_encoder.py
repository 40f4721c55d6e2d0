"""DynamicScatterVFE -- the pillar feature encoder of the mae_sst config.

Reference: mmdet3d/models/voxel_encoders/voxel_encoder.py:308-419 (+ DynamicVFE.__init__ :122-182,
DynamicVFELayer utils.py:107-144).  Same constructor arguments, same parameter names
(vfe_layers.{i}.linear.weight / .norm.*), same outputs (voxel_feats [V,128], voxel_coors [V,4] in
lexicographic (b,z,y,x) order).  The three torch.unique(dim=0) + torch_scatter calls per forward are
replaced by ONE pillar-segment build (usually shared with the detector, which passes `seg`) and the
segmented mean / max kernels of libgeomae_hip.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .registry import VOXEL_ENCODERS, build_norm_layer


class DynamicVFELayer(nn.Module):
    """Linear(no bias) -> norm -> ReLU (utils.py:107-144)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        return F.relu(self.norm(self.linear(inputs)))


@VOXEL_ENCODERS.register_module()
class DynamicScatterVFE(nn.Module):
    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, unique_once=False):
        super().__init__()
        assert mode in ["avg", "max"]
        assert len(feat_channels) > 0
        if fusion_layer is not None:
            raise NotImplementedError("fusion layers are not part of the pre-training path")
        if mode != "max":
            raise NotImplementedError("the mae_sst config pools with mode='max'")
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.vx, self.vy, self.vz = voxel_size
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        chans = [self.in_channels] + list(feat_channels)
        layers = []
        for i in range(len(chans) - 1):
            in_f = chans[i] * (2 if i > 0 else 1)
            layers.append(DynamicVFELayer(in_f, chans[i + 1], norm_cfg))
        self.vfe_layers = nn.ModuleList(layers)
        self.num_vfe = len(layers)
        self.rel_dist_scaler = rel_dist_scaler
        self.mode = mode
        self.unique_once = unique_once

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False, seg=None):
        """features [N, C_in] fp32, coors [N, 4] int32 (b, z, y, x)."""
        features = features.float()
        if seg is None:
            gx, gy, gz = ops.grid_size(self.voxel_size, self.point_cloud_range)
            batch_size = int(coors[:, 0].max().item()) + 1
            seg = ops.pillar_segment(coors.contiguous().int(), batch_size, (gz, gy, gx))
        V = seg.V
        inv = seg.inv.long()
        feats = [features]
        if self._with_cluster_center:
            with torch.no_grad():
                voxel_mean = ops.segment_mean_xyz(features.contiguous(), seg)
            f_cluster = features[:, :3] - voxel_mean[inv]
            feats.append(f_cluster / self.rel_dist_scaler if self.rel_dist_scaler != 1.0 else f_cluster)
        if self._with_voxel_center:
            c = coors.type_as(features)
            f_center = torch.stack([features[:, 0] - (c[:, 3] * self.vx + self.x_offset),
                                    features[:, 1] - (c[:, 2] * self.vy + self.y_offset),
                                    features[:, 2] - (c[:, 1] * self.vz + self.z_offset)], dim=1)
            feats.append(f_center)
        if self._with_distance:
            feats.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        x = torch.cat(feats, dim=-1)
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(x)
            voxel_feats = ops.segment_max(point_feats, seg, V)
            if i != len(self.vfe_layers) - 1:
                x = torch.cat([point_feats, voxel_feats[inv]], dim=1)
        if self.return_point_feats:
            return point_feats
        voxel_coors = seg.voxel_coors[:V]
        if return_inv:
            return voxel_feats, voxel_coors, inv
        return voxel_feats, voxel_coors
