"""Fine-tune path of the pre-trained SST encoder: DynamicVoxelNet + SSTInputLayer +
SSTSecondPretrainedv1 (+ SECONDFPN), the model of configs/pre_sst/m_sst_nus_second_pointpillar_fpn355_222_*.py up to
the multi-scale BEV features.

What is native here: dynamic voxelization and the fused DynamicScatterVFE (unchanged from pre-training), the
region-batching voxel drop (geomae_window_drop), the 6 encoder blocks through the same fused SST stack kernels
(window CSR instead of the reference's padded [W, 30|60|100, C] buckets + key-padding masks), and recover_bev
(geomae_recover_bev_*: channels-last canvas).  What is left to PyTorch / MIOpen, as SURVEY 8(f) prescribes: the
dense conv stack, SECONDFPN, and the detection head / assigner / NMS (mmdet3d's Anchor3DHead is not in this
image; `DynamicVoxelNet` takes any `bbox_head` module the caller registers).

Checkpoint interchange: `backbone.encoder_blocks.*` has the key names of the pre-training backbone
(MultiMAESSTSPChoose.encoder_blocks), which is what the reference's fine-tune run loads (`load_from` +
the commented loader at sst_second_pretrained_v1.py:88-99).
"""
import numpy as np
import torch
from torch import nn

from . import ops
from .registry import BACKBONES, DETECTORS, MIDDLE_ENCODERS, NECKS, MODELS, build_norm_layer
from .sst import BasicShiftBlock, PackedLayers, _FusedStack, pos_embed_table


def _drop_info_of(meta, training):
    """drop_info is either one dict or the (training, test) pair of the configs (sst_input_layer.py set_drop_info)."""
    if isinstance(meta, (tuple, list)):
        return meta[0] if training else meta[1]
    return meta


@MIDDLE_ENCODERS.register_module()
class SSTInputLayer(nn.Module):
    """mmdet3d/models/middle_encoders/sst_input_layer.py:15-107: regional grouping, voxel drop, batching info.
    Returns (voxel_feat_kept, layouts, voxel_info): `layouts` are the CSR window layouts of the kept voxels for the
    two shifts (the role of the reference's flat2win_inds_list), voxel_info carries `coors` (int64, as the backbone
    asserts), `voxel_keep_inds` and the per-shift drop levels."""

    def __init__(self, drop_info, shifts_list, window_shape, point_cloud_range, voxel_size, shuffle_voxels=True, debug=True):
        super().__init__()
        self.meta_drop_info = drop_info
        self.shifts_list, self.window_shape = shifts_list, tuple(window_shape)
        self.point_cloud_range, self.voxel_size = point_cloud_range, voxel_size
        self.shuffle_voxels, self.debug = shuffle_voxels, debug
        bev_x = int(np.ceil((point_cloud_range[3] - point_cloud_range[0]) / voxel_size[0]))
        bev_y = int(np.ceil((point_cloud_range[4] - point_cloud_range[1]) / voxel_size[1]))
        shift = shifts_list[1] if len(shifts_list) > 1 else (0, 0)
        self._wcfg = ops.make_window_config(self.window_shape, shift, (bev_x, bev_y))

    @torch.no_grad()
    def _keep_indices(self, coors, batch_size):
        """get_voxel_keep_inds (sst_input_layer.py:262-312): drop by shift 0, then by shift 1 among the survivors."""
        info = _drop_info_of(self.meta_drop_info, self.training)
        idx = torch.arange(coors.shape[0], device=coors.device)
        levels = []
        for s in range(len(self.shifts_list)):
            keep, lvl = ops.window_drop(coors[idx].contiguous(), batch_size, self._wcfg, s, info)
            levels = [l[keep] for l in levels] + [lvl[keep]]
            idx = idx[keep]
        return idx, levels

    def forward(self, voxel_feat, coors, batch_size):
        coors = coors.int()
        n = voxel_feat.shape[0]
        if self.shuffle_voxels:                                  # :70-78 (randperm; makes the drop uniform)
            perm = torch.randperm(n, device=voxel_feat.device)
            voxel_feat, coors = voxel_feat[perm], coors[perm]
        batch_size = int(batch_size)
        keep_inds, levels = self._keep_indices(coors.contiguous(), batch_size)
        voxel_feat, coors = voxel_feat[keep_inds], coors[keep_inds].contiguous()
        layouts = [ops.window_build(coors, batch_size, self._wcfg, s) for s in range(len(self.shifts_list))]
        voxel_info = dict(coors=coors.long(), voxel_keep_inds=keep_inds, batch_size=batch_size)
        for s, l in enumerate(levels):
            voxel_info[f"voxel_drop_level_shift{s}"] = l
        return voxel_feat, layouts, voxel_info


@BACKBONES.register_module()
class SSTSecondPretrainedv1(nn.Module):
    """mmdet3d/models/backbones/sst_second_pretrained_v1.py:18-241: (optional linear0) -> 6 BasicShiftBlocks ->
    recover_bev -> SECOND-style conv stages; returns the tuple of stage outputs."""

    def __init__(self, eval_flag=False, model_path="", d_model=[], nhead=[], num_blocks=6, dim_feedforward=[], dropout=0.0,
                 activation="gelu", output_shape=None, num_attached_conv=2, conv_in_channels=64,
                 conv_out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 norm_cfg=dict(type="naiveSyncBN2d", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False),
                 debug=True, drop_info=None, normalize_pos=False, pos_temperature=10000, window_shape=None, in_channel=None,
                 conv_kwargs=dict(kernel_size=3, dilation=2, padding=2, stride=1), checkpoint_blocks=[],
                 compute_dtype="bf16"):
        super().__init__()
        assert drop_info is not None and not normalize_pos
        assert len(set(d_model)) == 1 and d_model[0] == 128 and dim_feedforward[0] == 256, "kernels: d_model 128, ffn 256"
        self.meta_drop_info, self.pos_temperature, self.d_model = drop_info, pos_temperature, d_model
        self.window_shape, self.nhead, self.output_shape, self.debug = tuple(window_shape), nhead, output_shape, debug
        self.compute_dtype = compute_dtype
        if in_channel is not None:
            self.linear0 = nn.Linear(in_channel, d_model[0])
        self.encoder_blocks = nn.ModuleList([BasicShiftBlock(d_model[i], nhead[i], dim_feedforward[i], dropout, activation,
                                                             batch_first=False, block_id=i) for i in range(num_blocks)])
        self._reset_parameters()
        in_filters = [conv_in_channels, *conv_out_channels[:-1]]
        bias = bool(conv_cfg.get("bias", False))
        blocks = []
        for i, layer_num in enumerate(layer_nums):
            mods = [nn.Conv2d(in_filters[i], conv_out_channels[i], 3, stride=layer_strides[i], padding=1, bias=bias),
                    build_norm_layer(norm_cfg, conv_out_channels[i])[1], nn.ReLU(inplace=True)]
            for _ in range(layer_num):
                mods += [nn.Conv2d(conv_out_channels[i], conv_out_channels[i], 3, padding=1, bias=bias),
                         build_norm_layer(norm_cfg, conv_out_channels[i])[1], nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*mods))
        self.conv_blocks = nn.ModuleList(blocks)
        self.register_buffer("pos_table", pos_embed_table(self.window_shape, d_model[0], pos_temperature), persistent=False)
        self._packed = PackedLayers([l for b in self.encoder_blocks for l in b.encoder_list])

    def _reset_parameters(self):
        for name, p in self.named_parameters():                  # :237-240 (before the conv stack exists)
            if p.dim() > 1 and "scaler" not in name:
                nn.init.xavier_uniform_(p)

    @property
    def fused(self):
        return self.compute_dtype == "bf16"

    def forward(self, input_tuple):
        voxel_feat, layouts, voxel_info = input_tuple
        assert voxel_info["coors"].dtype == torch.int64, "data type of coors should be torch.int64!"
        batch_size = voxel_info.get("batch_size")
        if batch_size is None:
            batch_size = int(voxel_info["coors"][:, 0].max().item()) + 1
        x = voxel_feat.float()
        if hasattr(self, "linear0"):
            x = self.linear0(x)
        if self.fused:
            self._packed.refresh()
            x = _FusedStack.apply(x, self._packed, 0, 2 * len(self.encoder_blocks), layouts, self.pos_table, self.nhead[0])
        else:
            pos = [self.pos_table[L.tok_pos[:L.n].long()] for L in layouts]
            for block in self.encoder_blocks:
                x = block(x, pos, layouts, torch.float32)
        ny, nx = self.output_shape
        out = ops.recover_bev(x, voxel_info["coors"].int(), batch_size, ny, nx)
        outs = []
        for blk in self.conv_blocks:
            out = blk(out)
            outs.append(out)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    """mmdet3d/models/necks/second_fpn.py:11-95 (plain PyTorch; deconv / conv + norm + ReLU per level, concatenated)."""

    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False, init_cfg=None):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        deblocks = []
        for i, oc in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                up = nn.ConvTranspose2d(in_channels[i], oc, kernel_size=stride, stride=stride, bias=bool(upsample_cfg.get("bias", False)))
            else:
                k = int(np.round(1 / stride))
                up = nn.Conv2d(in_channels[i], oc, kernel_size=k, stride=k, bias=bool(conv_cfg.get("bias", False)))
            deblocks.append(nn.Sequential(up, build_norm_layer(norm_cfg, oc)[1], nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)
        for m in self.modules():                                  # init_cfg: Kaiming for the deconvs, BN weight 1
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


@DETECTORS.register_module()
class DynamicVoxelNet(nn.Module):
    """mmdet3d/models/detectors/dynamic_voxelnet.py:10-80 (VoxelNet with dynamic voxelization): voxelize -> voxel
    encoder -> middle encoder -> backbone -> neck.  `bbox_head` is built from the registry when its type is
    registered (mmdet3d's heads are not part of this package); forward_train then delegates to
    `bbox_head.forward_train(feats, img_metas, gt_bboxes_3d, gt_labels_3d, gt_bboxes_ignore)` if it exists, else
    `bbox_head(feats)` + `bbox_head.loss(...)` as the reference does (voxelnet.py, un-vendored base class)."""

    def __init__(self, voxel_layer, voxel_encoder, middle_encoder, backbone, centerpoint_head=False, neck=None,
                 bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        self.voxel_layer = ops.Voxelization(**voxel_layer)
        self.voxel_encoder = MODELS.build(voxel_encoder)
        self.middle_encoder = MODELS.build(middle_encoder)
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck) if neck is not None else None
        self.bbox_head = None
        if bbox_head is not None and isinstance(bbox_head, dict) and bbox_head.get("type") in MODELS:
            self.bbox_head = MODELS.build(dict(bbox_head, train_cfg=train_cfg, test_cfg=test_cfg))
        elif isinstance(bbox_head, nn.Module):
            self.bbox_head = bbox_head
        self.centerpoint_head, self.train_cfg, self.test_cfg = centerpoint_head, train_cfg, test_cfg
        vs, rng = self.voxel_layer.voxel_size, self.voxel_layer.point_cloud_range
        gx, gy, gz = ops.grid_size(vs, rng)
        self.grid_size = (gz, gy, gx)

    @property
    def with_neck(self):
        return self.neck is not None

    @torch.no_grad()
    def voxelize(self, points):
        """:57-80: per-sample dynamic voxelization + batch index; here one kernel over the concatenated batch."""
        sizes = [int(p.shape[0]) for p in points]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        pts = torch.cat([p.float() for p in points], dim=0).contiguous() if len(points) > 1 else points[0].float().contiguous()
        boffs = torch.from_numpy(offs).pin_memory().to(pts.device, non_blocking=True)
        vs = self.voxel_layer.voxel_size
        top, _, _ = ops.voxelize_batch3(pts, boffs, len(points), vs, vs, vs, self.voxel_layer.point_cloud_range)
        return pts, top

    def extract_feat(self, points, img_metas=None):
        voxels, coors = self.voxelize(points)
        seg = ops.pillar_segment(coors, len(points), self.grid_size)
        voxel_features, feature_coors = self.voxel_encoder(voxels, coors, seg=seg)
        batch_size = len(points)                                  # == coors[-1, 0] + 1 (:49) without the readback
        x = self.middle_encoder(voxel_features, feature_coors, batch_size)
        x = self.backbone(x)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_train(self, points, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_bboxes_ignore=None):
        x = self.extract_feat(points, img_metas)
        if self.bbox_head is None:
            raise RuntimeError("DynamicVoxelNet.forward_train needs a bbox_head (register one in geomae_amd.registry.HEADS); "
                               "extract_feat() returns the neck features")
        if hasattr(self.bbox_head, "forward_train"):
            return self.bbox_head.forward_train(x, img_metas, gt_bboxes_3d, gt_labels_3d, gt_bboxes_ignore)
        outs = self.bbox_head(x)
        return self.bbox_head.loss(*outs, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.extract_feat(kwargs["points"], kwargs.get("img_metas"))
