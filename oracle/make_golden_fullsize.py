"""Generate tests/golden/g_fullsize.npz: SUMMARY fixtures (losses + per-parameter gradient norms + a few small gradient
tensors) of the reference's pre-training step at the FULL sizes of BASELINE.json configs 2, 3 and 4 (build container
only; TEST INFRASTRUCTURE ONLY; data only).
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_fullsize.py
  c2   : 4 single-sweep frames (~26 k points each), the mae_sst model as is (6 + 2 + 2 blocks)
  c3   : one 10-sweep frame (~260 k points)
  c4   : one Waymo-geometry frame (range +-74.88 x [-2, 4] m, 0.32 m pillars, grid 468^2:
         configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:8-10; 64 beams, ~180 k points)
Same procedure as oracle/make_golden.py::pipeline (the reference's own modules under the stubs of ref_import.py, its
compiled C++ voxelizer); frames are regenerated from seeds by the tests (geomae_amd.synth)."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import ref_import  # noqa: E402
import geomae_oracle as O  # noqa: E402
from geomae_amd import synth  # noqa: E402

torch.set_num_threads(8)
ref = ref_import.load_reference()

sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize_cases import CASES, CASES_C1, make_frames  # noqa: E402

ALL = dict(CASES, **CASES_C1)


def frames_of(case):
    return make_frames(ALL[case][1])


def run(case):
    geo, _ = ALL[case]
    frames = frames_of(case)
    RANGE, B = geo["range"], len(frames)
    ny, nx = geo["grid"][1:]
    enc, dec = geo.get("blocks", (6, 2))
    params = O.make_params(7, enc, dec)
    vfe = ref.vfe.DynamicScatterVFE(in_channels=5, feat_channels=[64, 128], with_distance=False, voxel_size=geo["top"],
                                    with_cluster_center=True, with_voxel_center=True, point_cloud_range=RANGE,
                                    norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01))
    drop_info = ({0: {"max_tokens": 56, "drop_range": (0, 56)}, 1: {"max_tokens": 144, "drop_range": (56, 100000)}},) * 2
    bb = ref.bb.MultiMAESSTSPChoose(cls_sub_voxel=True, window_shape=(12, 12), shifts_list=[(0, 0), (6, 6)],
                                    point_cloud_range=RANGE, voxel_size=geo["top"], shuffle_voxels=False, low=False,
                                    med=False, top=True, d_model=[128] * 6, nhead=[8] * 6, sub_voxel_ratio_low=(8, 4, 4),
                                    sub_voxel_ratio_med=(4, 2, 2), encoder_num_blocks=enc, decoder_num_blocks=dec,
                                    dim_feedforward=[256] * 6, output_shape=[ny, nx], debug=True, drop_info=drop_info,
                                    pos_temperature=10000, normalize_pos=False)
    vfe.load_state_dict({k[len("voxel_encoder."):]: v for k, v in params.items() if k.startswith("voxel_encoder.")}, strict=False)
    bb.load_state_dict({k[len("backbone."):]: v for k, v in params.items() if k.startswith("backbone.")}, strict=True)
    vfe.train(); bb.train()
    model_cfg = dict(grid_size=geo["grid"], sub_voxel_ratio_low=(8, 4, 4), sub_voxel_ratio_med=(4, 2, 2),
                     voxel_layer=dict(voxel_size=geo["top"], point_cloud_range=RANGE),
                     sub_voxel_layer_low=dict(voxel_size=geo["low"]), sub_voxel_layer_med=dict(voxel_size=geo["med"]),
                     random_mask_ratio=0.7, spatial_shape=list(geo["grid"]), mse_loss=True, cls_sub_voxel=True,
                     loss_ratio_low=10.0, loss_ratio_med=8.0, loss_ratio_top=10.0, loss_ratio_low_nor=4.0,
                     cls_loss_ratio_low=5.0, cls_loss_ratio_med=2.0, normalize_sub_voxel=True)
    det = ref_import.make_detector(ref, model_cfg)

    def vox(vs):
        cs = []
        for i, p in enumerate(frames):
            pts = torch.as_tensor(p)
            c = pts.new_zeros((pts.shape[0], 3), dtype=torch.int32)
            ref.voxel_layer.dynamic_voxelize(pts, c, list(map(float, vs)), list(map(float, RANGE)), 3)
            cs.append(F.pad(c, (1, 0), value=i))
        return torch.cat(cs, 0)
    t0 = time.time()
    voxels = torch.cat([torch.as_tensor(p) for p in frames], 0)
    coors, sub_low, sub_med = vox(geo["top"]), vox(geo["low"]), vox(geo["med"])
    voxel_features, feature_coors = vfe(voxels, coors)
    g = torch.Generator().manual_seed(5)
    ids_keep, ids_mask = O.vanilla_mask_index(feature_coors.numpy(), B, 0.7, g)
    ik, im = torch.as_tensor(ids_keep), torch.as_tensor(ids_mask)
    c_low, vc_low, _ = det.get_centroid_per_voxel(voxels[:, [2, 1, 0]], sub_low)
    c_med, vc_med, _ = det.get_centroid_per_voxel(voxels[:, [2, 1, 0]], sub_med)
    c_top, vc_top, _ = det.get_centroid_per_voxel(voxels[:, [2, 1, 0]], coors)
    med_curv, med_curv_mask = det.get_multi_voxel_id_to_tensor_id_for_curv(feature_coors.long(), vc_med.long(), c_med, B)
    pair = ref_import._indice_pairs_subm_3x3(feature_coors, B, list(geo["grid"]), ksize=[1, 3, 3])[2]
    normal, curv = det.cal_regular_voxel_nor_and_curv(med_curv, med_curv_mask, c_top, pair.long())
    nc_low = det.normalize_centroid_sub_voxel(vc_low[:, 1:], c_low, layer="low")
    nc_med = det.normalize_centroid_sub_voxel(vc_med[:, 1:], c_med, layer="med")
    nc_top = det.normalize_centroid_sub_voxel(vc_top[:, 1:], c_top, layer="top")
    t_low, m_low, t_med, m_med = det.get_multi_voxel_id_to_tensor_id_ori(feature_coors.long(), vc_low.long(), vc_med.long(),
                                                                        nc_low, nc_med, im, B)
    normal_canon = O.canonical_sign(normal)
    x = bb(voxel_features[ik], feature_coors[ik], feature_coors[im], B)
    reg_low, reg_med, reg_top, _, _, nor_top, cls_low, cls_med = x
    loss = det.forward_loss(t_low, m_low, t_med, m_med, nc_top[im], normal_canon[im], None, None, reg_low, reg_med, reg_top,
                            None, None, nor_top, cls_low, cls_med)
    sum(loss.values()).backward()
    named = {"voxel_encoder." + k: v for k, v in vfe.named_parameters()}
    named.update({"backbone." + k: v for k, v in bb.named_parameters()})
    out = {f"{case}.n_points": np.array([p.shape[0] for p in frames], np.int64), f"{case}.V": np.int64(feature_coors.shape[0]),
           f"{case}.ids_keep": ids_keep.astype(np.int32), f"{case}.ids_mask": ids_mask.astype(np.int32),
           f"{case}.coors_checksum": np.int64(feature_coors.long().sum()),
           f"{case}.loss_names": np.array(list(loss.keys())), f"{case}.loss_vals": np.array([float(v) for v in loss.values()]),
           f"{case}.grad_names": np.array(list(named.keys())),
           f"{case}.grad_norms": np.array([float(v.grad.double().norm()) for v in named.values()]),
           f"{case}.grad_vfe0": named["voxel_encoder.vfe_layers.0.linear.weight"].grad.numpy(),
           f"{case}.grad_mask_token": named["backbone.mask_token"].grad.numpy(),
           f"{case}.grad_pred_top_w": named["backbone.decoder_pred_top.weight"].grad.numpy(),
           f"{case}.grad_enc5_ffn_b": named[f"backbone.encoder_blocks.{enc - 1}.encoder_list.1.linear1.bias"].grad.numpy(),
           f"{case}.grad_dec_out_w": named[f"backbone.decoder_centroid_blocks.{dec - 1}.encoder_list.1.win_attn.self_attn.out_proj.weight"].grad.numpy(),
           f"{case}.voxel_coors": feature_coors.numpy().astype(np.int16)}
    if case in CASES:
        del out[f"{case}.voxel_coors"]                    # (full-size cases keep the checksum only)
    print(case, "N", voxels.shape[0], "V", feature_coors.shape[0], "M", len(ids_mask), "losses",
          {k: round(float(v), 5) for k, v in loss.items()}, f"{time.time() - t0:.1f} s", flush=True)
    return out


def main():
    """no argument: the three full-size cases -> g_fullsize.npz; `c1`: BASELINE config 1's geometry (0.5 m pillars, grid
    205, SST-tiny 1 + 1 blocks) on the 16 k uniform cloud, a LiDAR-ring cloud and both as a batch -> g_pipeline_c1.npz"""
    args = sys.argv[1:]
    c1 = "c1" in args
    which = list(CASES_C1) if c1 else ([a for a in args if a in CASES] or list(CASES))
    dst = os.path.join(os.environ.get("GEOMAE_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")),
                       "g_pipeline_c1.npz" if c1 else "g_fullsize.npz")
    out = dict(np.load(dst)) if os.path.exists(dst) else {}
    for case in which:
        out.update(run(case))
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
