"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

TEST INFRASTRUCTURE ONLY.  The reference's Python is imported from /root/reference under
stub modules (oracle/ref_import.py) and its C++ voxelizer is the compiled oracle/_ref; only
inputs/outputs (data) are written -- no reference source text.  Re-run:
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
(GEOMAE_GOLDEN_OUT=<dir> writes there instead of tests/golden/; oracle/verify_golden.py uses it.)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import ref_import                      # noqa: E402
import geomae_oracle as O              # noqa: E402
from geomae_amd import synth           # noqa: E402

# GEOMAE_GOLDEN_OUT: scratch directory for a verification run (oracle/verify_golden.py) -- the committed fixtures stay untouched
OUT = os.environ.get("GEOMAE_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
ref = ref_import.load_reference()

RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
LEVELS = dict(top=(0.256, 0.256, 8), med=(0.128, 0.128, 2), low=(0.064, 0.064, 1))


def ref_voxelize(points, vs, rng=RANGE):
    pts = torch.as_tensor(points)
    coors = pts.new_zeros((pts.shape[0], 3), dtype=torch.int32)
    ref.voxel_layer.dynamic_voxelize(pts, coors, list(map(float, vs)), list(map(float, rng)), 3)
    return coors.numpy()


def g1_voxelize():
    out = {}
    clouds = dict(uniform=synth.uniform_cloud(0, 16000), boundary=synth.boundary_cloud(),
                  lidar=synth.lidar_frame(2))
    # config-1 voxel sizes too (0.5 / 0.25 / 0.125 m)
    levels = dict(LEVELS, c1_top=(0.5, 0.5, 8), c1_med=(0.25, 0.25, 2), c1_low=(0.125, 0.125, 1))
    for cname, pts in clouds.items():
        if cname == "boundary":
            out["boundary_points"] = pts
        out[f"{cname}_n"] = np.int64(pts.shape[0])
        out[f"{cname}_xyzsum"] = pts[:, :3].astype(np.float64).sum(0)
        for lname, vs in levels.items():
            out[f"{cname}_{lname}"] = ref_voxelize(pts, vs).astype(np.int16)
    # Waymo-like geometry (config 4)
    wr = [-74.88, -74.88, -2.0, 74.88, 74.88, 4.0]
    pts = synth.uniform_cloud(4, 8000, wr)
    out["waymo_n"] = np.int64(pts.shape[0])
    for lname, vs in dict(top=(0.32, 0.32, 6), med=(0.16, 0.16, 1.5), low=(0.08, 0.08, 0.75)).items():
        out[f"waymo_{lname}"] = ref_voxelize(pts, vs, wr).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "g1_voxelize.npz"), **out)
    print("g1", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def small_scene():
    return [synth.lidar_frame(11, beams=16, n_az=400), synth.lidar_frame(12, beams=16, n_az=360)]


def pipeline(enc_blocks, dec_blocks, tag, frames):
    cfg = O.mae_sst_cfg(enc_blocks, dec_blocks)
    B = len(frames)
    params = O.make_params(7, enc_blocks, dec_blocks)
    # ---- reference modules with these weights
    vfe = ref.vfe.DynamicScatterVFE(in_channels=5, feat_channels=[64, 128], with_distance=False,
                                    voxel_size=LEVELS["top"], with_cluster_center=True, with_voxel_center=True,
                                    point_cloud_range=RANGE,
                                    norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01))
    drop_info = ({0: {"max_tokens": 56, "drop_range": (0, 56)}, 1: {"max_tokens": 144, "drop_range": (56, 100000)}},) * 2
    bb = ref.bb.MultiMAESSTSPChoose(cls_sub_voxel=True, window_shape=(12, 12), shifts_list=[(0, 0), (6, 6)],
                                    point_cloud_range=RANGE, voxel_size=LEVELS["top"], shuffle_voxels=False,
                                    low=False, med=False, top=True, d_model=[128] * 6, nhead=[8] * 6,
                                    sub_voxel_ratio_low=(8, 4, 4), sub_voxel_ratio_med=(4, 2, 2),
                                    encoder_num_blocks=enc_blocks, decoder_num_blocks=dec_blocks,
                                    dim_feedforward=[256] * 6, output_shape=[400, 400], debug=True,
                                    drop_info=drop_info, pos_temperature=10000, normalize_pos=False)
    vsd = {k[len("voxel_encoder."):]: v for k, v in params.items() if k.startswith("voxel_encoder.")}
    missing = vfe.load_state_dict(vsd, strict=False)
    assert not missing.unexpected_keys and all("running" in k or "num_batches" in k for k in missing.missing_keys)
    bsd = {k[len("backbone."):]: v for k, v in params.items() if k.startswith("backbone.")}
    bb.load_state_dict(bsd, strict=True)
    vfe.train(); bb.train()
    for p_ in list(vfe.parameters()) + list(bb.parameters()):
        p_.requires_grad_(True)
    model_cfg = dict(grid_size=(1, 400, 400), sub_voxel_ratio_low=(8, 4, 4), sub_voxel_ratio_med=(4, 2, 2),
                     voxel_layer=dict(voxel_size=LEVELS["top"], point_cloud_range=RANGE),
                     sub_voxel_layer_low=dict(voxel_size=LEVELS["low"]), sub_voxel_layer_med=dict(voxel_size=LEVELS["med"]),
                     random_mask_ratio=0.7, spatial_shape=[1, 400, 400], mse_loss=True, cls_sub_voxel=True,
                     loss_ratio_low=10.0, loss_ratio_med=8.0, loss_ratio_top=10.0, loss_ratio_low_nor=4.0,
                     cls_loss_ratio_low=5.0, cls_loss_ratio_med=2.0, normalize_sub_voxel=True)
    det = ref_import.make_detector(ref, model_cfg)

    # ---- extract_feat, line by line (ssl.py:169-242), calling the reference's own methods
    def vox(vs):
        cs = [F.pad(torch.as_tensor(ref_voxelize(p, vs)), (1, 0), value=i) for i, p in enumerate(frames)]
        return torch.cat(cs, 0)
    voxels = torch.cat([torch.as_tensor(p) for p in frames], 0)
    coors, sub_low, sub_med = vox(LEVELS["top"]), vox(LEVELS["low"]), vox(LEVELS["med"])
    voxel_features, feature_coors = vfe(voxels, coors)
    g = torch.Generator().manual_seed(5)
    ids_keep, ids_mask = O.vanilla_mask_index(feature_coors.numpy(), B, 0.7, g)
    ids_keep_t, ids_mask_t = torch.as_tensor(ids_keep), torch.as_tensor(ids_mask)
    c_low, vc_low, n_low = det.get_centroid_per_voxel(voxels[:, [2, 1, 0]], sub_low)
    c_med, vc_med, n_med = det.get_centroid_per_voxel(voxels[:, [2, 1, 0]], sub_med)
    c_top, vc_top, n_top = det.get_centroid_per_voxel(voxels[:, [2, 1, 0]], coors)
    med_curv, med_curv_mask = det.get_multi_voxel_id_to_tensor_id_for_curv(feature_coors.long(), vc_med.long(), c_med, B)
    pair = ref_import._indice_pairs_subm_3x3(feature_coors, B, [1, 400, 400], ksize=[1, 3, 3])[2]
    normal, curv = det.cal_regular_voxel_nor_and_curv(med_curv, med_curv_mask, c_top, pair.long())
    nc_low = det.normalize_centroid_sub_voxel(vc_low[:, 1:], c_low, layer="low")
    nc_med = det.normalize_centroid_sub_voxel(vc_med[:, 1:], c_med, layer="med")
    nc_top = det.normalize_centroid_sub_voxel(vc_top[:, 1:], c_top, layer="top")
    t_low, m_low, t_med, m_med = det.get_multi_voxel_id_to_tensor_id_ori(
        feature_coors.long(), vc_low.long(), vc_med.long(), nc_low, nc_med, ids_mask_t, B)
    t_top = nc_top[ids_mask_t]
    mask_coors = feature_coors[ids_mask_t]
    # the sign of torch.svd's vector is backend-defined; the fixture stores the raw reference
    # normal AND the loss computed with the build's canonical sign rule (documented in DESIGN.md)
    normal_canon = O.canonical_sign(normal)
    x = bb(voxel_features[ids_keep_t], feature_coors[ids_keep_t], mask_coors, B)
    reg_low, reg_med, reg_top, _, _, nor_top, cls_low, cls_med = x
    loss = det.forward_loss(t_low, m_low, t_med, m_med, t_top, normal_canon[ids_mask_t], None, None,
                            reg_low, reg_med, reg_top, None, None, nor_top, cls_low, cls_med)
    total = sum(loss.values())
    total.backward()
    named = {"voxel_encoder." + k: v for k, v in vfe.named_parameters()}
    named.update({"backbone." + k: v for k, v in bb.named_parameters()})
    out = dict(
        seeds=np.array([11, 12]), n_points=np.array([p.shape[0] for p in frames]),
        coors_top=coors.numpy().astype(np.int16), coors_med=sub_med.numpy().astype(np.int16),
        coors_low=sub_low.numpy().astype(np.int16),
        voxel_coors=feature_coors.numpy().astype(np.int16),
        voxel_feats=voxel_features.detach().numpy(),
        ids_keep=ids_keep.astype(np.int32), ids_mask=ids_mask.astype(np.int32),
        vc_low=vc_low.numpy().astype(np.int16), vc_med=vc_med.numpy().astype(np.int16),
        c_low=c_low.numpy(), c_med=c_med.numpy(), c_top=c_top.numpy(),
        n_low=n_low.numpy().astype(np.int32), n_med=n_med.numpy().astype(np.int32), n_top=n_top.numpy().astype(np.int32),
        med_curv_mask=np.packbits(med_curv_mask.numpy()), pair=pair.numpy(),
        normal=normal.numpy(), curv=curv.numpy(),
        t_low_mask=np.packbits(m_low.numpy()), t_med_mask=np.packbits(m_med.numpy()),
        t_low_vals=t_low[m_low].numpy(), t_med_vals=t_med[m_med].numpy(), t_top=t_top.numpy(),
        reg_top=reg_top.detach().numpy(), nor_top=nor_top.detach().numpy(),
        reg_med=reg_med.detach().numpy(), cls_med=cls_med.detach().numpy(),
        reg_low_sum=reg_low.detach().double().sum(0).numpy(), cls_low_sum=cls_low.detach().double().sum(0).numpy(),
        loss_names=np.array(list(loss.keys())), loss_vals=np.array([float(v) for v in loss.values()], np.float64),
    )
    # window plumbing of the encoder pass (bb.py:143-196): ids and in-window coords per shift
    info = {}
    info = bb.window_partition(feature_coors[ids_keep_t].long(), info)
    for s in (0, 1):
        out[f"enc_win_s{s}"] = info[f"batch_win_inds_shift{s}"].numpy().astype(np.int32)
        out[f"enc_ciw_s{s}"] = info[f"coors_in_win_shift{s}"].numpy().astype(np.int8)
    out["pos_table"] = bb.get_pos_embed.__func__ is not None and O.pos_embed_table((12, 12), 128).numpy()  # replaced below
    # real pos-embed from the reference for every (cx, cy): call with identity layout
    ciw = torch.stack([torch.arange(12).repeat_interleave(12), torch.arange(12).repeat(12)], -1)
    lvl = torch.zeros(144, dtype=torch.long)
    bb.drop_info = {0: {"max_tokens": 144, "drop_range": (0, 100000)}}
    ind = {0: (torch.arange(144), (torch.arange(144),))}
    out["pos_table"] = bb.get_pos_embed(ind, ciw, lvl, torch.float32, None)[0].reshape(144, 128).numpy()
    # gradients: norms for every parameter + a few full small tensors
    gn = np.array([float(v.grad.double().norm()) if v.grad is not None else -1.0 for v in named.values()])
    out["grad_names"] = np.array(list(named.keys()))
    out["grad_norms"] = gn
    out["grad_vfe0"] = named["voxel_encoder.vfe_layers.0.linear.weight"].grad.numpy()
    out["grad_mask_token"] = named["backbone.mask_token"].grad.numpy()
    out["grad_pred_top_w"] = named["backbone.decoder_pred_top.weight"].grad.numpy()
    out["grad_enc0_inproj_bias"] = named["backbone.encoder_blocks.0.encoder_list.0.win_attn.self_attn.in_proj_bias"].grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"g_pipeline_{tag}.npz"), **out)
    print(tag, "V", feature_coors.shape[0], "N", voxels.shape[0], "losses", {k: round(float(v), 5) for k, v in loss.items()})


def g_hard_voxelize():
    """voxel_layer.hard_voxelize of the reference (CPU path, voxelization_cpu.cpp:42-132) on seeded clouds:
    a LiDAR frame with the PointPillars-style config, a dense cloud that hits max_points and max_voxels, and a
    cloud with out-of-range points (clamped into the border cells by this fork)."""
    out = {}
    cases = dict(
        lidar=(synth.lidar_frame(5, beams=16, n_az=600), (0.25, 0.25, 8.0), RANGE, 20, 30000),
        dense=(synth.uniform_cloud(6, 20000, [-4, -4, -1, 4, 4, 1]), (0.5, 0.5, 0.5), [-4.0, -4.0, -1.0, 4.0, 4.0, 1.0], 8, 600),
        clamp=(synth.uniform_cloud(7, 6000, [-12, -12, -3, 12, 12, 3]), (1.0, 1.0, 2.0), [-8.0, -8.0, -2.0, 8.0, 8.0, 2.0], 5, 200))
    for name, (pts, vs, rng, mp, mv) in cases.items():
        p = torch.as_tensor(pts)
        voxels = p.new_zeros((mv, mp, p.shape[1]))
        coors = p.new_zeros((mv, 3), dtype=torch.int32)
        num = p.new_zeros((mv,), dtype=torch.int32)
        n = ref.voxel_layer.hard_voxelize(p, voxels, coors, num, list(map(float, vs)), list(map(float, rng)), mp, mv, 3)
        out[f"{name}_cfg"] = np.array([*vs, *rng, mp, mv], dtype=np.float64)
        out[f"{name}_points"] = pts if name != "lidar" else np.zeros(0, np.float32)     # lidar: regenerated by seed
        out[f"{name}_voxel_num"] = np.int64(n)
        out[f"{name}_coors"] = coors[:n].numpy().astype(np.int16)
        out[f"{name}_num"] = num[:n].numpy().astype(np.int16)
        # the kept point rows are identified by their xyz sum per voxel (full voxels would be large)
        out[f"{name}_voxel_sum"] = voxels[:n].numpy().astype(np.float64).sum(axis=(1, 2))
        out[f"{name}_first_voxels"] = voxels[:16].numpy()
        print("hard", name, "voxels", n, "max pts", int(num.max()))
    np.savez_compressed(os.path.join(OUT, "g_hard_voxelize.npz"), **out)


FT_DROP = {0: dict(max_tokens=30, drop_range=(0, 30)), 1: dict(max_tokens=60, drop_range=(30, 60)),
           2: dict(max_tokens=144, drop_range=(60, 100000))}
FT_CFG = dict(d_model=[128, 128], nhead=[8, 8], num_blocks=1, dim_feedforward=[256, 256], output_shape=[400, 400], conv_in_channels=128,
              conv_out_channels=[32, 48], layer_nums=[1, 2], layer_strides=[2, 2], debug=False, drop_info=(FT_DROP, FT_DROP),
              pos_temperature=10000, normalize_pos=False, window_shape=(12, 12))


def g_finetune():
    """SSTInputLayer + SSTSecondPretrainedv1 of the reference (sst_input_layer.py, sst_second_pretrained_v1.py) on a
    small scene, fp32, training mode, no shuffle; the bucket sizes are chosen so that nothing is dropped (what is dropped
    in an over-full window is random in the reference).  Outputs are summarised (the maps are 200x200 and 100x100)."""
    frames = [synth.lidar_frame(31, beams=16, n_az=500), synth.lidar_frame(32, beams=16, n_az=400)]
    _, coors = O.voxelize_batch(frames, LEVELS["top"], RANGE)
    vc = O.unique_rows(coors)[0]
    n = vc.shape[0]
    feat = torch.randn(n, 128, generator=torch.Generator().manual_seed(3))
    mid = ref.mid.SSTInputLayer(drop_info=(FT_DROP, FT_DROP), shifts_list=[(0, 0), (6, 6)], window_shape=(12, 12),
                                point_cloud_range=RANGE, voxel_size=LEVELS["top"], shuffle_voxels=False, debug=False)
    bb = ref.ft.SSTSecondPretrainedv1(**FT_CFG)
    state = O.seeded_state(5, {k: v.shape for k, v in bb.state_dict().items()})
    bb.load_state_dict(state)
    mid.train()
    bb.train()
    x = feat.clone().requires_grad_(True)
    out_tuple = mid(x, torch.as_tensor(vc).long(), 2)
    assert out_tuple[0].shape[0] == n                                  # nothing dropped
    outs = bb(out_tuple)
    w = [torch.randn(o.shape, generator=torch.Generator().manual_seed(11 + i)) for i, o in enumerate(outs)]
    loss = sum((o * wi).sum() for o, wi in zip(outs, w)) * 1e-2
    loss.backward()
    res = dict(coors=vc.astype(np.int16), n=np.int64(n), loss=np.float64(loss.item()), dx=x.grad.numpy())
    for i, o in enumerate(outs):
        o = o.detach()
        res[f"out{i}_shape"] = np.array(o.shape)
        res[f"out{i}_sum"] = np.float64(o.double().sum())
        res[f"out{i}_abs"] = np.float64(o.double().abs().sum())
        res[f"out{i}_chan"] = o.double().sum(dim=(0, 2, 3)).numpy()
        res[f"out{i}_patch"] = o[:, :, 96:104, 96:104].numpy()
    gn = {k: float(p.grad.double().norm()) for k, p in bb.named_parameters()}
    res["grad_names"] = np.array(sorted(gn))
    res["grad_norms"] = np.array([gn[k] for k in sorted(gn)])
    np.savez_compressed(os.path.join(OUT, "g_finetune.npz"), **res)
    print("finetune n", n, "loss", float(loss), {f"out{i}": tuple(o.shape) for i, o in enumerate(outs)})


if __name__ == "__main__":
    if "--finetune-only" in sys.argv:
        g_finetune()
        sys.exit(0)
    if "--hard-only" in sys.argv:
        g_hard_voxelize()
        sys.exit(0)
    g1_voxelize()
    g_hard_voxelize()
    g_finetune()
    pipeline(1, 1, "tiny", small_scene())
    pipeline(6, 2, "full", small_scene())
