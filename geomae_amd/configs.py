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


# optimizer / schedule of configs/_base_/schedules/cosine_2x.py:1-17 (AdamW, 'norm' params undecayed, clip 10)
OPTIMIZER = dict(type="AdamW", lr=1e-5, betas=(0.9, 0.999), weight_decay=0.05,
                 paramwise_cfg=dict(custom_keys={"norm": dict(decay_mult=0.0)}))
GRAD_CLIP = dict(max_norm=10, norm_type=2)
SAMPLES_PER_GPU = 4
