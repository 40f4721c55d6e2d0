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


class _FusedVFE(torch.autograd.Function):
    """The whole encoder as seven sweeps forward / three backward (geomae_amd/csrc/vfe.hip).  The six
    parameters enter as autograd inputs only so that the node is part of the graph; their gradients are
    accumulated straight into .grad by the kernels."""

    @staticmethod
    def forward(ctx, points, enc, seg, w0, g0, b0, w1, g1, b1):
        from torch import distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        sync = world > 1 and isinstance(enc.vfe_layers[0].norm, __import__("geomae_amd").norm.NaiveSyncBatchNorm1d)
        world = world if sync else 1
        plan = ops.VfePlan(points, seg, w0, w1, (enc.vx, enc.vy, enc.vz), (enc.x_offset, enc.y_offset, enc.z_offset),
                           layer1_bf16=enc.layer1_bf16)
        vf, m0 = ops.vfe_forward(plan, enc.vfe_layers[0].norm, enc.vfe_layers[1].norm, world)
        ctx.plan, ctx.world = plan, world
        ctx.params = dict(w0=w0, g0=g0, b0=b0, w1=w1, g1=g1, b1=b1)
        ctx.save_for_backward(m0, vf)
        return vf

    @staticmethod
    def backward(ctx, dvf):
        m0, vf = ctx.saved_tensors
        ops.vfe_backward(ctx.plan, m0, vf, dvf, ctx.params, ctx.world)
        return (None,) * 9


def _vfe_world(enc):
    from torch import distributed as dist
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    sync = world > 1 and isinstance(enc.vfe_layers[0].norm, __import__("geomae_amd").norm.NaiveSyncBatchNorm1d)
    return world if sync else 1


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

    use_fused = True      # set False to run the composed (ATen + segment kernels) form, kept for A/B tests
    # Arithmetic of the fused sweeps' two 128 x 128 layer-1 GEMMs (utils.py:130-144's Linear): 'fp32' = bf16 x 3 split products
    # (fp32 grade: what a stand-alone encoder and the tight parity tests run), 'bf16' = one bf16 MFMA product with fp32
    # accumulation, as SURVEY 8(d) lists this GEMM for BASELINE config 2.  Not a constructor argument (the reference's config
    # has none): the detector copies its backbone's compute_dtype here, so the whole step runs in one mode.
    compute_dtype = "fp32"

    @property
    def layer1_bf16(self):
        return self.compute_dtype == "bf16"

    def fused_ok(self, features, seg, return_inv):
        return (self.use_fused and seg is not None and self.training and features.is_cuda and not return_inv
                and not self.return_point_feats and features.shape[1] == 5 and self._with_cluster_center
                and self._with_voxel_center and not self._with_distance and self.rel_dist_scaler == 1.0
                and [l.linear.out_features for l in self.vfe_layers] == [64, 128])

    # ---- explicit (autograd-free) schedule of the fused path: detector.train_step_explicit
    @torch.no_grad()
    def prepare_points(self, features, seg):
        """Weight-independent front of the fused path (pillar means, decorated features in pillar order): may run
        ahead of time, e.g. with the next batch's voxelization (detector.prefetch)."""
        return ops.vfe_prepare_points(features, seg, (self.vx, self.vy, self.vz), (self.x_offset, self.y_offset, self.z_offset))

    @torch.no_grad()
    def forward_explicit(self, features, seg, zeros=None, prepared=None):
        """zeros: optional ops.ZeroArena sized by ops.vfe_forward_zero_specs(seg.cap, seg.V, prepared is not None): the
        sweeps' accumulator buffers come out of it and the library skips its own memsets.  prepared: result of
        prepare_points for this batch."""
        l0, l1 = self.vfe_layers
        world = _vfe_world(self)
        import contextlib
        with (ops.prezeroed() if zeros is not None else contextlib.nullcontext()):
            plan = ops.VfePlan(features, seg, l0.linear.weight, l1.linear.weight, (self.vx, self.vy, self.vz),
                               (self.x_offset, self.y_offset, self.z_offset), zeros=zeros, prepared=prepared,
                               layer1_bf16=self.layer1_bf16)
            vf, m0 = ops.vfe_forward(plan, l0.norm, l1.norm, world, zeros=zeros)
        return vf, (plan, m0, vf, world)

    @torch.no_grad()
    def backward_explicit(self, state, dvf, zeros=None, side=None):
        plan, m0, vf, world = state
        l0, l1 = self.vfe_layers
        import contextlib
        with (ops.prezeroed() if zeros is not None else contextlib.nullcontext()):
            ops.vfe_backward(plan, m0, vf, dvf, dict(w0=l0.linear.weight, g0=l0.norm.weight, b0=l0.norm.bias,
                                                     w1=l1.linear.weight, g1=l1.norm.weight, b1=l1.norm.bias), world,
                             zeros=zeros, side=side)

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False, seg=None):
        """features [N, C_in] fp32, coors [N, 4] int32 (b, z, y, x)."""
        features = features.float()
        if self.fused_ok(features, seg, return_inv):
            l0, l1 = self.vfe_layers
            vf = _FusedVFE.apply(features.contiguous(), self, seg, l0.linear.weight, l0.norm.weight, l0.norm.bias,
                                 l1.linear.weight, l1.norm.weight, l1.norm.bias)
            return vf, seg.voxel_coors[:seg.V]
        if seg is None:
            gx, gy, gz = ops.grid_size(self.voxel_size, self.point_cloud_range)
            batch_size = int(coors[:, 0].max().item()) + 1
            seg = ops.pillar_segment(coors.contiguous().int(), batch_size, (gz, gy, gx))
        V = seg.V
        inv = seg.inv.long()
        if bool((inv < 0).any()):
            # pillar_segment marks points outside the grid with inv = -1 (csrc/segment.hip); the fused path never sees
            # them (they are not in `order`).  Indexing with -1 would wrap to the last pillar: refuse instead
            raise RuntimeError("DynamicScatterVFE (composed path): points outside the voxel grid; filter them first "
                               "(PointsRangeFilter) or use the fused path")
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
