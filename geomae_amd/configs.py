"""The `model = dict(...)` of the reference's mae_sst pre-training config, restated as data.

Values follow configs/mae_sst/m_sst_nus_singlestage_curv_07_ssl_dataset_wo_dbsampler_6x_1e-5.py:14-161
(tests/test_config_cpu.py checks this dict against the reference file when it is mounted).  Users
normally load their own config file with geomae_amd.Config.fromfile; bench.py / smoke() use this
one because /root/reference does not exist on the GPU box.
"""


def mae_sst_model(encoder_num_blocks=6, decoder_num_blocks=2, voxel_size=(0.256, 0.256, 8),
                  sub_voxel_size_low=(0.064, 0.064, 1), sub_voxel_size_med=(0.128, 0.128, 2),
                  point_cloud_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), grid_size=(1, 400, 400)):
    point_cloud_range = list(point_cloud_range)
    window_shape = (12, 12)
    sub_voxel_ratio_low, sub_voxel_ratio_med = (8, 4, 4), (4, 2, 2)
    drop_info_training = {0: {"max_tokens": 56, "drop_range": (0, 56)},
                          1: {"max_tokens": 144, "drop_range": (56, 100000)}}
    drop_info_test = {0: {"max_tokens": 32, "drop_range": (0, 32)}, 1: {"max_tokens": 72, "drop_range": (32, 72)},
                      2: {"max_tokens": 144, "drop_range": (72, 100000)}}

    def vl(size, max_num_points=-1, max_voxels=(-1, -1)):
        return dict(voxel_size=size, max_num_points=max_num_points, point_cloud_range=point_cloud_range,
                    max_voxels=max_voxels)

    return dict(
        type="MultiSubVoxelDynamicVoxelNetSSL", normalize_sub_voxel=True, mse_loss=True,
        loss=dict(type="SmoothL1Loss", reduction="mean", loss_weight=1.0),
        spatial_shape=list(grid_size), loss_ratio_low_nor=4.0, loss_ratio_med_nor=0, loss_ratio_top_nor=0,
        loss_ratio_low=10.0, loss_ratio_med=8.0, loss_ratio_top=10.0, cls_sub_voxel=True, cls_loss_ratio_low=5.0,
        cls_loss_ratio_med=2.0, random_mask_ratio=0.7, grid_size=tuple(grid_size),
        sub_voxel_ratio_low=sub_voxel_ratio_low, sub_voxel_ratio_med=sub_voxel_ratio_med,
        voxel_layer=vl(voxel_size), sub_voxel_layer_low=vl(sub_voxel_size_low),
        sub_voxel_layer_med=vl(sub_voxel_size_med),
        hard_sub_voxel_layer_low=vl(sub_voxel_size_low, 30, (140000, 140000)),
        hard_sub_voxel_layer_med=vl(sub_voxel_size_med, 50, (80000, 80000)),
        hard_sub_voxel_layer_top=vl(voxel_size, 100, (40000, 40000)),
        voxel_encoder=dict(type="DynamicScatterVFE", in_channels=5, feat_channels=[64, 128], with_distance=False,
                           voxel_size=voxel_size, with_cluster_center=True, with_voxel_center=True,
                           point_cloud_range=point_cloud_range,
                           norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)),
        backbone=dict(type="MultiMAESSTSPChoose", cls_sub_voxel=True, window_shape=window_shape,
                      shifts_list=[(0, 0), (window_shape[0] // 2, window_shape[1] // 2)],
                      point_cloud_range=point_cloud_range, voxel_size=voxel_size, shuffle_voxels=False, low=False,
                      med=False, top=True, d_model=[128] * 6, nhead=[8] * 6,
                      sub_voxel_ratio_low=sub_voxel_ratio_low, sub_voxel_ratio_med=sub_voxel_ratio_med,
                      encoder_num_blocks=encoder_num_blocks, decoder_num_blocks=decoder_num_blocks,
                      dim_feedforward=[256] * 6, output_shape=[400, 400], debug=True,
                      drop_info=(drop_info_training, drop_info_test), pos_temperature=10000, normalize_pos=False))


def pre_sst_model(num_blocks=6, conv_out_channels=(128, 128, 256), layer_nums=(3, 5, 5)):
    """The merged `model = dict(...)` of the fine-tune config (BASELINE config 5): configs/_base_/models/sst_base_nus.py
    + configs/pre_sst/m_sst_nus_second_pointpillar_fpn355_222_curv_07_ssl_data_wo_dbsampler_6x_1e-5.py:60-163, as data
    (tests/test_config_cpu.py compares it with the reference file when that is mounted)."""
    voxel_size, window_shape = (0.25, 0.25, 8), (12, 12)
    pcr = [-50, -50, -5.0, 50, 50, 3.0]
    lvl = {0: {"max_tokens": 32, "drop_range": (0, 32)}, 1: {"max_tokens": 72, "drop_range": (32, 72)},
           2: {"max_tokens": 144, "drop_range": (72, 1000)}}
    drop_info = (lvl, {k: dict(v) for k, v in lvl.items()})
    z = [-1.80032795, -1.74440365, -1.68526504, -1.67339111, -1.61785072, -1.80984986, -1.763965]
    return dict(
        type="DynamicVoxelNet", centerpoint_head=False,
        voxel_layer=dict(voxel_size=voxel_size, max_num_points=-1, point_cloud_range=pcr, max_voxels=(-1, -1)),
        voxel_encoder=dict(type="DynamicScatterVFE", in_channels=5, feat_channels=[64, 128], with_distance=False,
                           voxel_size=voxel_size, with_cluster_center=True, with_voxel_center=True, point_cloud_range=pcr,
                           norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)),
        middle_encoder=dict(type="SSTInputLayer", window_shape=window_shape, shifts_list=[(0, 0), (6, 6)],
                            point_cloud_range=pcr, voxel_size=voxel_size, shuffle_voxels=True, debug=True, drop_info=drop_info),
        backbone=dict(type="SSTSecondPretrainedv1", d_model=[128] * 6, nhead=[8] * 6, num_blocks=num_blocks,
                      dim_feedforward=[256] * 6, output_shape=[400, 400], conv_in_channels=128,
                      conv_out_channels=list(conv_out_channels), layer_nums=list(layer_nums), layer_strides=[2, 2, 2],
                      debug=True, drop_info=drop_info, pos_temperature=10000, normalize_pos=False, window_shape=window_shape,
                      eval_flag=False, model_path=""),
        neck=dict(type="SECONDFPN", norm_cfg=dict(type="naiveSyncBN2d", eps=1e-3, momentum=0.01),
                  in_channels=list(conv_out_channels), upsample_strides=[1, 2, 4], out_channels=[128, 128, 128]),
        bbox_head=dict(
            type="Anchor3DHead", num_classes=10, in_channels=384, feat_channels=384, use_direction_classifier=True,
            anchor_generator=dict(
                type="AlignedAnchor3DRangeGenerator", ranges=[[-49.6, -49.6, v, 49.6, 49.6, v] for v in z],
                sizes=[[4.60718145, 1.95017717, 1.72270761], [6.73778078, 2.4560939, 2.73004906],
                       [12.01320693, 2.87427237, 3.81509561], [1.68452161, 0.60058911, 1.27192197],
                       [0.7256437, 0.66344886, 1.75748069], [0.40359262, 0.39694519, 1.06232151],
                       [0.48578221, 2.49008838, 0.98297065]],
                custom_values=[0, 0], rotations=[0, 1.57], reshape_out=True),
            assigner_per_size=False, diff_rad_by_sin=True, dir_offset=-0.7854,
            bbox_coder=dict(type="DeltaXYZWLHRBBoxCoder", code_size=9),
            loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
            loss_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0),
            loss_dir=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.2)),
        train_cfg=dict(assigner=dict(type="MaxIoUAssigner", iou_calculator=dict(type="BboxOverlapsNearest3D"), pos_iou_thr=0.6,
                                     neg_iou_thr=0.3, min_pos_iou=0.3, ignore_iof_thr=-1),
                       allowed_border=0, code_weight=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2], pos_weight=-1, debug=False,
                       point_cloud_range=pcr),
        test_cfg=dict(use_rotate_nms=True, nms_across_levels=False, nms_pre=1000, nms_thr=0.2, score_thr=0.05,
                      min_bbox_size=0, max_num=500, pts=dict(pc_range=pcr[:2])))


# optimizer / schedule of configs/_base_/schedules/cosine_2x.py:1-17 (AdamW, 'norm' params undecayed, clip 10)
OPTIMIZER = dict(type="AdamW", lr=1e-5, betas=(0.9, 0.999), weight_decay=0.05,
                 paramwise_cfg=dict(custom_keys={"norm": dict(decay_mult=0.0)}))
GRAD_CLIP = dict(max_norm=10, norm_type=2)
SAMPLES_PER_GPU = 4
