"""CPU oracle: a restatement of the reference's GeoMAE-SST pre-training hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (geomae_amd), which fails loudly without
its HIP library.  Plain numpy / torch-CPU fp32, written from the reference's behaviour;
each function cites the reference lines it follows (paths relative to /root/reference).
"ssl.py" = mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py,
"bb.py"  = mmdet3d/models/backbones/multi_mae_sst_spearate_top_only.py.

Pinning (see tests/test_oracle_golden.py): every function here is checked against
fixtures under tests/golden/ that were produced by importing the reference itself
(oracle/make_golden.py; voxelizer = the reference's C++ compiled as oracle/_ref).
Three un-vendored dependencies are restated from their published behaviour and are
"parity unpinned" by the reference: torch_scatter (segment mean/max), spconv 2.1.21
sub-manifold 3x3 index pairs, mmdet 2.20 CrossEntropyLoss(use_sigmoid=True).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# A1  dynamic voxelization   (mmdet3d/ops/voxel/src/voxelization_cpu.cpp:6-40,139-168;
#                             CUDA twin voxelization_cuda.cu:22-63,375-377)
# --------------------------------------------------------------------------------------


def grid_size_f32(voxel_size, pc_range):
    """grid[j] = (int)ceil((max_j - min_j) / vs_j) evaluated in fp32 (voxelization_cpu.cpp:154-157)."""
    vs = np.asarray(voxel_size, np.float32)
    r = np.asarray(pc_range, np.float32)
    return np.ceil((r[3:] - r[:3]) / vs).astype(np.int32)  # (x, y, z)


def dynamic_voxelize(points, voxel_size, pc_range):
    """coors[i] = (cz, cy, cx), c_d = clamp(floor((p_d - min_d) / vs_d), 0, grid_d - 1); all fp32.

    This fork clamps out-of-range points into the border cell (voxelization_cpu.cpp:22-31).
    NaN follows the C cast used by the reference's CPU build ((int)NaN == INT_MIN -> 0).
    """
    p = np.asarray(points, np.float32)[:, :3]
    vs = np.asarray(voxel_size, np.float32)
    lo = np.asarray(pc_range, np.float32)[:3]
    grid = grid_size_f32(voxel_size, pc_range)
    q = np.floor((p - lo) / vs)                      # fp32 sub, IEEE fp32 divide, floor
    with np.errstate(invalid="ignore"):
        q = np.where(np.isnan(q), -1.0, q)
        c = np.clip(q, -1.0, grid.astype(np.float32)).astype(np.int64)
    c = np.clip(c, 0, grid - 1).astype(np.int32)
    return np.ascontiguousarray(c[:, ::-1])          # (z, y, x)


def voxelize_batch(points_list, voxel_size, pc_range):
    """ssl.py:355-377: per-sample voxelize, prepend batch index, concatenate."""
    coors = []
    for b, p in enumerate(points_list):
        c = dynamic_voxelize(p, voxel_size, pc_range)
        coors.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], axis=1))
    return np.concatenate([np.asarray(p, np.float32) for p in points_list], 0), np.concatenate(coors, 0)


# --------------------------------------------------------------------------------------
# A2  scatter_v2  (mmdet3d/ops/sst/sst_ops.py:8-39) -- torch.unique(dim=0) + torch_scatter
# --------------------------------------------------------------------------------------


def unique_rows(coors):
    """torch.unique(coors, dim=0, return_inverse, return_counts): lexicographic row order."""
    c = np.asarray(coors)
    u, inv, cnt = np.unique(c, axis=0, return_inverse=True, return_counts=True)
    return u, inv.reshape(-1).astype(np.int64), cnt.astype(np.int64)


def segment_mean(feat, inv, n):
    out = torch.zeros((n, feat.shape[1]), dtype=feat.dtype)
    out = out.index_add(0, inv, feat)
    cnt = torch.bincount(inv, minlength=n).to(feat.dtype).clamp(min=1)
    return out / cnt[:, None]


def segment_max(feat, inv, n):
    """torch_scatter.scatter_max values (differentiable: gradient goes to the arg-max rows)."""
    idx = inv[:, None].expand_as(feat)
    out = torch.full((n, feat.shape[1]), float("-inf"), dtype=feat.dtype)
    return out.scatter_reduce(0, idx, feat, reduce="amax", include_self=True)


def scatter_v2(feat, coors, mode, unq=None):
    new_coors, inv, _ = unq if unq is not None else unique_rows(coors)
    inv_t = torch.as_tensor(inv)
    n = new_coors.shape[0]
    if mode in ("avg", "mean"):
        return segment_mean(feat, inv_t, n), new_coors, inv_t
    if mode == "max":
        return segment_max(feat, inv_t, n), new_coors, inv_t
    raise NotImplementedError(mode)


# --------------------------------------------------------------------------------------
# A3/A4  DynamicScatterVFE  (mmdet3d/models/voxel_encoders/voxel_encoder.py:358-419,
#        DynamicVFELayer utils.py:130-144, naiveSyncBN1d ops/norm.py:54-86)
# --------------------------------------------------------------------------------------


def batch_norm_train(x, weight, bias, eps, stats=None):
    """Training-mode BN over dim 0; `stats` = (mean, meansqr) when supplied by the SyncBN exchange
    (ops/norm.py:64-80: var = E[x^2] - E[x]^2).  Single-process = nn.BatchNorm1d (biased var)."""
    if stats is None:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
    else:
        mean, meansqr = stats
        var = meansqr - mean * mean
    invstd = torch.rsqrt(var + eps)
    return (x - mean) * (invstd * weight) + bias


def vfe_forward(params, points, coors, voxel_size, pc_range, eps=1e-3, prefix="voxel_encoder.",
                unq=None):
    """points [N,5] fp32 tensor, coors [N,4] int (b,z,y,x) -> voxel_feats [V,128], voxel_coors [V,4].

    Feature decoration order (voxel_encoder.py:372-397): raw(5) | xyz - cluster mean(3) |
    xyz - pillar centre(3), centre = coor * v + (v/2 + range_min) with the offset a Python double
    rounded to fp32 when it meets the fp32 tensor (voxel_encoder.py:152-157,385-390).
    """
    unq = unq if unq is not None else unique_rows(coors)
    new_coors, inv, _ = unq
    inv_t = torch.as_tensor(inv)
    V = new_coors.shape[0]
    xyz = points[:, :3]
    mean = segment_mean(xyz, inv_t, V)
    f_cluster = xyz - mean[inv_t]
    c = torch.as_tensor(np.asarray(coors)).to(points.dtype)
    vx, vy, vz = voxel_size
    xo, yo, zo = vx / 2 + pc_range[0], vy / 2 + pc_range[1], vz / 2 + pc_range[2]
    f_center = torch.stack([xyz[:, 0] - (c[:, 3] * vx + xo),
                            xyz[:, 1] - (c[:, 2] * vy + yo),
                            xyz[:, 2] - (c[:, 1] * vz + zo)], dim=1)
    feats = torch.cat([points, f_cluster, f_center], dim=1)
    n_layers = 2
    for i in range(n_layers):
        w = params[f"{prefix}vfe_layers.{i}.linear.weight"]
        g = params[f"{prefix}vfe_layers.{i}.norm.weight"]
        b = params[f"{prefix}vfe_layers.{i}.norm.bias"]
        x = F.linear(feats, w)
        x = batch_norm_train(x, g, b, eps)
        pf = F.relu(x)
        vf = segment_max(pf, inv_t, V)
        if i != n_layers - 1:
            feats = torch.cat([pf, vf[inv_t]], dim=1)
    return vf, new_coors, inv_t


# --------------------------------------------------------------------------------------
# A5  centroids per (sub-)voxel   (ssl.py:726-768) ; points are passed as (z, y, x)
# --------------------------------------------------------------------------------------


def centroid_per_voxel(points_zyx, coors):
    u, inv, cnt = unique_rows(coors)
    inv_t = torch.as_tensor(inv)
    s = torch.zeros((u.shape[0], 3), dtype=torch.float32).index_add(0, inv_t, points_zyx)
    return s / torch.as_tensor(cnt).float()[:, None], u, cnt


# --------------------------------------------------------------------------------------
# A6  random masking  (ssl.py:287-304)
# --------------------------------------------------------------------------------------


def vanilla_mask_index(voxel_coors, batch_size, mask_ratio, generator):
    keep, mask = [], []
    b = np.asarray(voxel_coors)[:, 0]
    for i in range(batch_size):
        inds = np.nonzero(b == i)[0]
        L = inds.shape[0]
        len_keep = int(L * (1 - mask_ratio))
        perm = torch.randperm(L, generator=generator).numpy()
        keep.append(inds[perm[:len_keep]])
        mask.append(inds[perm[len_keep:]])
    return np.concatenate(keep), np.concatenate(mask)


# --------------------------------------------------------------------------------------
# A7/A11  dense per-pillar sub-voxel tables  (ssl.py:643-722)
# --------------------------------------------------------------------------------------


def _parent_and_slot(voxel_coors, sub_coors, ratio, grid_size, batch_size):
    gz, gy, gx = grid_size
    grid_shape = gz * gy * gx
    vc = np.asarray(voxel_coors).astype(np.int64)
    sc = np.asarray(sub_coors).astype(np.int64)
    table = np.zeros(batch_size * grid_shape, np.int64)       # zeros, as the reference (ssl.py:654)
    table[vc[:, 0] * grid_shape + vc[:, 2] * gy + vc[:, 3]] = np.arange(vc.shape[0])
    parent = table[sc[:, 0] * grid_shape + (sc[:, 2] // ratio[1]) * gy + sc[:, 3] // ratio[2]]
    slot = (sc[:, 1] % ratio[0]) * (ratio[1] * ratio[2]) + (sc[:, 2] % ratio[1]) * ratio[2] + sc[:, 3] % ratio[2]
    return parent, slot


def dense_sub_voxel(voxel_coors, sub_coors, sub_centroids, ratio, grid_size, batch_size):
    """-> ([V, S, 3] fp32, [V, S] bool), S = prod(ratio); slot = (z%rz)*ry*rx + (y%ry)*rx + x%rx."""
    V = np.asarray(voxel_coors).shape[0]
    S = ratio[0] * ratio[1] * ratio[2]
    parent, slot = _parent_and_slot(voxel_coors, sub_coors, ratio, grid_size, batch_size)
    tgt = torch.zeros((V * S, 3), dtype=torch.float32)
    msk = torch.zeros((V * S,), dtype=torch.bool)
    flat = torch.as_tensor(parent * S + slot)
    tgt[flat] = sub_centroids
    msk[flat] = True
    return tgt.view(V, S, 3), msk.view(V, S)


# --------------------------------------------------------------------------------------
# A8  3x3 BEV sub-manifold neighbour table (spconv 2.1.21 stand-in; call site ssl.py:192-207)
# --------------------------------------------------------------------------------------


def neighbour_pairs_3x3(voxel_coors, batch_size, spatial_shape):
    """pair[k, i], k = (dy+1)*3 + (dx+1): row of the pillar at (y+dy, x+dx) of pillar i, or -1."""
    _, ny, nx = spatial_shape
    vc = np.asarray(voxel_coors).astype(np.int64)
    V = vc.shape[0]
    table = np.full(batch_size * ny * nx, -1, np.int64)
    table[vc[:, 0] * ny * nx + vc[:, 2] * nx + vc[:, 3]] = np.arange(V)
    pair = np.full((9, V), -1, np.int64)
    k = 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            y, x = vc[:, 2] + dy, vc[:, 3] + dx
            ok = (y >= 0) & (y < ny) & (x >= 0) & (x < nx)
            lin = vc[:, 0] * ny * nx + np.clip(y, 0, ny - 1) * nx + np.clip(x, 0, nx - 1)
            pair[k] = np.where(ok, table[lin], -1)
            k += 1
    return pair


# --------------------------------------------------------------------------------------
# A9  normal + curvature from the 3x3 neighbourhood covariance  (ssl.py:575-610)
# --------------------------------------------------------------------------------------


def canonical_sign(normal):
    """The build's deterministic sign rule (the reference leaves the SVD sign to the backend,
    ssl.py:598-602): flip so that the component of largest magnitude is positive (ties: lowest
    index).  Applied identically by the HIP kernel."""
    a = normal.abs()
    k = a.argmax(dim=-1, keepdim=True)
    s = torch.gather(normal, -1, k)
    return torch.where(s < 0, -normal, normal)


def scatter_matrix(dense_med, dense_med_mask, top_centroid, pair):
    """cov[i] = sum over valid (neighbour k, slot s) of d d^T, d = med_centroid - top_centroid[i]."""
    pair_t = torch.as_tensor(pair)
    absent = pair_t == -1
    around = dense_med[pair_t].clone()                 # [9, V, S, 3]   (index -1 wraps, then zeroed)
    around_mask = dense_med_mask[pair_t].clone()
    around[absent] = 0
    around_mask[absent] = False
    V = dense_med.shape[0]
    around = around.transpose(0, 1).contiguous().view(V, -1, 3)
    around_mask = around_mask.transpose(0, 1).contiguous().view(V, -1)
    own = top_centroid[:, None, :].repeat(1, around.shape[1], 1)
    own[~around_mask] = 0
    d = around - own
    return d.transpose(-2, -1) @ d, around_mask.sum(1)


def normal_and_curv(dense_med, dense_med_mask, top_centroid, pair, canonical=False):
    cov, npts = scatter_matrix(dense_med, dense_med_mask, top_centroid, pair)
    U, S, Vh = torch.linalg.svd(cov)                    # torch.svd(cov)[2] == Vh^T
    normal = Vh[..., -1, :]
    normal = normal / torch.norm(normal, p=2, dim=-1, keepdim=True)
    if canonical:
        normal = canonical_sign(normal)
    curv = S.to(torch.float64) + 1e-9
    curv = curv / curv.sum(dim=-1, keepdim=True)
    return normal, curv, cov, S, npts


# --------------------------------------------------------------------------------------
# A10  cell-local normalisation  (ssl.py:626-641)
# --------------------------------------------------------------------------------------


def normalize_centroid(coors_zyx, centroids, cell_size_xyz, pc_range):
    vs = torch.tensor(list(cell_size_xyz)[::-1], dtype=torch.float32)
    start = torch.tensor(list(pc_range[:3])[::-1], dtype=torch.float32)
    origin = torch.as_tensor(np.asarray(coors_zyx)) * vs + start      # int * fp32 -> fp32
    return (centroids - origin) / vs


# --------------------------------------------------------------------------------------
# A12-A15  window partition / buckets / in-window order  (bb.py:413-681)
# --------------------------------------------------------------------------------------


def window_partition(coors, window_shape, shifts_list, voxel_size, pc_range):
    """-> per shift: batch_win_inds [n], coors_in_win [n,2] (x, y)   (bb.py:628-659)."""
    c = np.asarray(coors).astype(np.int64)
    wx, wy = window_shape
    bev_x = int(np.ceil((pc_range[3] - pc_range[0]) / voxel_size[0]))
    bev_y = int(np.ceil((pc_range[4] - pc_range[1]) / voxel_size[1]))
    nwx = int(np.ceil(bev_x / wx) + 1)
    nwy = int(np.ceil(bev_y / wy) + 1)
    out = []
    for sx, sy in shifts_list:
        x = c[:, 3] + (wx - sx if sx > 0 else 0)
        y = c[:, 2] + (wy - sy if sy > 0 else 0)
        win = c[:, 0] * (nwx * nwy) + (x // wx) * nwy + y // wy
        out.append((win, np.stack([x % wx, y % wy], axis=-1)))
    return out, nwx * nwy


def window_layout(win_inds, drop_info):
    """Bucket ("drop level") per token and a padded slot for each token (bb.py:413-454,519-541).

    Returns {level: (window_ids_sorted [W], token_index [W, T] (-1 = padding))} plus the
    keep mask (all True when every window has <= its bucket's max_tokens, as in training).
    The order of tokens inside a window is the stable order of their flat index; the reference's
    is whatever torch.sort returns (bb.py:468) -- attention is invariant to it.
    """
    win = np.asarray(win_inds)
    uniq, inv, cnt = np.unique(win, return_inverse=True, return_counts=True)
    order = np.argsort(inv, kind="stable")
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    inner = np.empty_like(win)
    inner[order] = np.arange(win.shape[0]) - np.repeat(starts, cnt)
    n_per_tok = cnt[inv]
    level = np.full(win.shape, -1, np.int64)
    tgt = np.zeros(win.shape, np.int64)
    for dl, info in drop_info.items():
        lo, hi = info["drop_range"]
        m = (n_per_tok >= lo) & (n_per_tok < hi)
        level[m] = dl
        tgt[m] = info["max_tokens"]
    keep = inner < tgt
    layout = {}
    for dl, info in drop_info.items():
        T = info["max_tokens"]
        wsel = np.nonzero((cnt >= info["drop_range"][0]) & (cnt < info["drop_range"][1]))[0]
        if wsel.size == 0:
            continue
        remap = np.full(uniq.shape[0], -1, np.int64)
        remap[wsel] = np.arange(wsel.size)
        tok = np.nonzero((level == dl) & keep)[0]
        table = np.full((wsel.size, T), -1, np.int64)
        table[remap[inv[tok]], inner[tok]] = tok
        layout[dl] = (uniq[wsel], table)
    return layout, level, keep


def pos_embed_table(window_shape, d_model, temperature=10000):
    """[wx*wy, d_model] table indexed cx*wy + cy  (bb.py:361-394), fp32 like the reference."""
    wx, wy = window_shape
    cx = torch.arange(wx, dtype=torch.float32).repeat_interleave(wy)
    cy = torch.arange(wy, dtype=torch.float32).repeat(wx)
    return pos_embed(torch.stack([cx, cy], -1), window_shape, d_model, temperature)


def pos_embed(coors_in_win, window_shape, d_model, temperature=10000):
    wx, wy = window_shape
    c = torch.as_tensor(np.asarray(coors_in_win))
    x = c[:, 0] - wx / 2
    y = c[:, 1] - wy / 2
    pos_length = d_model // 2
    inv_freq = torch.arange(pos_length, dtype=torch.float32)
    inv_freq = temperature ** (2 * (inv_freq // 2) / pos_length)
    ex = x[:, None] / inv_freq[None, :]
    ey = y[:, None] / inv_freq[None, :]
    ex = torch.stack([ex[:, ::2].sin(), ex[:, 1::2].cos()], dim=-1).flatten(1)
    ey = torch.stack([ey[:, ::2].sin(), ey[:, 1::2].cos()], dim=-1).flatten(1)
    return torch.cat([ex, ey], dim=-1).to(torch.float32)


# --------------------------------------------------------------------------------------
# A16-A20  windowed attention layer  (mmdet3d/models/sst/sst_basic_block.py:26-147)
# --------------------------------------------------------------------------------------


def window_attention(x, pos, layout, p, prefix, nhead):
    """Padded per-bucket multi-head attention == nn.MultiheadAttention(q=k=x+pos, v=x,
    key_padding_mask) of sst_basic_block.py:36-59, restated with explicit matmuls."""
    C = x.shape[1]
    dh = C // nhead
    w_in, b_in = p[prefix + "self_attn.in_proj_weight"], p[prefix + "self_attn.in_proj_bias"]
    w_o, b_o = p[prefix + "self_attn.out_proj.weight"], p[prefix + "self_attn.out_proj.bias"]
    out = torch.zeros_like(x)
    for dl, (_, table) in layout.items():
        tab = torch.as_tensor(table)
        pad = tab < 0
        idx = tab.clamp(min=0)
        xw = x[idx] * (~pad)[..., None]                 # zero padded rows (flat2window)
        pw = pos[idx] * (~pad)[..., None]
        W, T = tab.shape
        qk_in = xw + pw
        q = F.linear(qk_in, w_in[:C], b_in[:C])
        k = F.linear(qk_in, w_in[C:2 * C], b_in[C:2 * C])
        v = F.linear(xw, w_in[2 * C:], b_in[2 * C:])
        q = q.view(W, T, nhead, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
        k = k.view(W, T, nhead, dh).transpose(1, 2)
        v = v.view(W, T, nhead, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
        a = torch.softmax(s, dim=-1)
        o = (a @ v).transpose(1, 2).reshape(W, T, C)
        o = F.linear(o, w_o, b_o)
        sel = ~pad
        out = out.index_put((tab[sel],), o[sel])
    return out


def encoder_layer(x, pos, layout, p, prefix, nhead):
    """Post-norm layer (sst_basic_block.py:85-102), dropout 0, GELU (erf)."""
    a = window_attention(x, pos, layout, p, prefix + "win_attn.", nhead)
    x = F.layer_norm(x + a, (x.shape[1],), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])
    h = F.linear(F.gelu(F.linear(x, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"])),
                 p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])
    return F.layer_norm(x + h, (x.shape[1],), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"])


def shift_block(x, pos_list, layout_list, p, prefix, nhead):
    for i in range(2):
        s = i % len(layout_list)
        x = encoder_layer(x, pos_list[s], layout_list[s], p, f"{prefix}encoder_list.{i}.", nhead)
    return x


def voxel_info(coors, cfg):
    parts, _ = window_partition(coors, cfg["window_shape"], cfg["shifts_list"], cfg["voxel_size"],
                                cfg["point_cloud_range"])
    layouts, pos = [], []
    for win, ciw in parts:
        layout, _, keep = window_layout(win, cfg["drop_info"])
        assert keep.all(), "token drop is not part of the training path (SURVEY 3.3)"
        layouts.append(layout)
        pos.append(pos_embed(ciw, cfg["window_shape"], cfg["d_model"], cfg.get("pos_temperature", 10000)))
    return layouts, pos


# --------------------------------------------------------------------------------------
# A21/A22  backbone  (bb.py:136-303)
# --------------------------------------------------------------------------------------


def backbone_forward(p, voxel_feat, coors, coors_mask, cfg, prefix="backbone."):
    nhead = cfg["nhead"]
    layouts, pos = voxel_info(coors, cfg)
    x = voxel_feat
    for i in range(cfg["encoder_num_blocks"]):
        x = shift_block(x, pos, layouts, p, f"{prefix}encoder_blocks.{i}.", nhead)
    n_vis = coors.shape[0]
    M = coors_mask.shape[0]
    tokens = torch.cat([x, p[prefix + "mask_token"].repeat(M, 1)], dim=0)
    coors_all = np.concatenate([np.asarray(coors), np.asarray(coors_mask)], axis=0)
    layouts, pos = voxel_info(coors_all, cfg)
    cen, den = tokens, tokens
    for i in range(cfg["decoder_num_blocks"]):
        cen = shift_block(cen, pos, layouts, p, f"{prefix}decoder_centroid_blocks.{i}.", nhead)
    for i in range(cfg["decoder_num_blocks"]):
        den = shift_block(den, pos, layouts, p, f"{prefix}decoder_density_blocks.{i}.", nhead)
    cm, dm = cen[n_vis:], den[n_vis:]

    def head(name, t):
        return F.linear(t, p[f"{prefix}{name}.weight"], p[f"{prefix}{name}.bias"])

    s_low = cfg["per_sub_voxel_num_low"]
    s_med = cfg["per_sub_voxel_num_med"]
    return dict(reg_low=head("decoder_pred_low", cm).view(-1, s_low, 3),
                reg_med=head("decoder_pred_med", cm).view(-1, s_med, 3),
                reg_top=head("decoder_pred_top", cm),
                nor_top=head("decoder_pred_density_top", dm),
                cls_low=head("cls_pred_low", cm).view(-1, s_low, 2),
                cls_med=head("cls_pred_med", cm).view(-1, s_med, 2),
                encoded=x, dec_centroid=cen, dec_density=den)


# --------------------------------------------------------------------------------------
# A23  losses  (ssl.py:837-902; BCE = mmdet CrossEntropyLoss(use_sigmoid=True) restated)
# --------------------------------------------------------------------------------------


def sigmoid_ce(pred_k2, label_k):
    onehot = F.one_hot(label_k, 2).to(pred_k2.dtype)
    return F.binary_cross_entropy_with_logits(pred_k2, onehot, reduction="mean")


def forward_loss(pred, tgt, w):
    ml = tgt["mask_low"].reshape(-1)
    mm = tgt["mask_med"].reshape(-1)

    def mse(a, b, ratio):
        l = ((a - b) ** 2).mean(dim=-1)
        return l.sum() / l.shape[0] * ratio

    return dict(
        loss_curv_around=mse(pred["nor_top"], tgt["normal"], w["loss_ratio_low_nor"]),
        loss_centroid_low=mse(pred["reg_low"].reshape(-1, 3)[ml], tgt["centroid_low"].reshape(-1, 3)[ml],
                              w["loss_ratio_low"]),
        loss_centroid_med=mse(pred["reg_med"].reshape(-1, 3)[mm], tgt["centroid_med"].reshape(-1, 3)[mm],
                              w["loss_ratio_med"]),
        loss_centroid_top=mse(pred["reg_top"], tgt["centroid_top"], w["loss_ratio_top"]),
        loss_cls_low=sigmoid_ce(pred["cls_low"].reshape(-1, 2), ml.long()) * w["cls_loss_ratio_low"],
        loss_cls_med=sigmoid_ce(pred["cls_med"].reshape(-1, 2), mm.long()) * w["cls_loss_ratio_med"],
    )


# --------------------------------------------------------------------------------------
# A24  forward_train / extract_feat orchestration  (ssl.py:126-242)
# --------------------------------------------------------------------------------------


def geometric_targets(points, coors_top, coors_med, coors_low, voxel_coors, cfg, batch_size,
                      canonical=True):
    """Everything extract_feat computes between the VFE and the backbone (ssl.py:185-219)."""
    pz = points[:, [2, 1, 0]]
    c_low, vc_low, _ = centroid_per_voxel(pz, coors_low)
    c_med, vc_med, _ = centroid_per_voxel(pz, coors_med)
    c_top, vc_top, _ = centroid_per_voxel(pz, coors_top)
    assert np.array_equal(vc_top, np.asarray(voxel_coors))
    grid = cfg["grid_size"]
    med_raw, med_raw_mask = dense_sub_voxel(voxel_coors, vc_med, c_med, cfg["sub_voxel_ratio_med"], grid, batch_size)
    pair = neighbour_pairs_3x3(voxel_coors, batch_size, cfg["spatial_shape"])
    normal, curv, cov, S, npts = normal_and_curv(med_raw, med_raw_mask, c_top, pair, canonical=canonical)
    rng = cfg["point_cloud_range"]
    n_low = normalize_centroid(vc_low[:, 1:], c_low, cfg["sub_voxel_size_low"], rng)
    n_med = normalize_centroid(vc_med[:, 1:], c_med, cfg["sub_voxel_size_med"], rng)
    n_top = normalize_centroid(vc_top[:, 1:], c_top, cfg["voxel_size"], rng)
    d_low, m_low = dense_sub_voxel(voxel_coors, vc_low, n_low, cfg["sub_voxel_ratio_low"], grid, batch_size)
    d_med, m_med = dense_sub_voxel(voxel_coors, vc_med, n_med, cfg["sub_voxel_ratio_med"], grid, batch_size)
    return dict(centroid_low=d_low, mask_low=m_low, centroid_med=d_med, mask_med=m_med,
                centroid_top=n_top, normal=normal, curv=curv, cov=cov, sing=S, npts=npts,
                med_raw=med_raw, med_raw_mask=med_raw_mask, pair=pair, top_raw=c_top)


def forward_train(params, points_list, cfg, ids_keep=None, ids_mask=None, generator=None,
                  canonical=True):
    """points_list: list of [N_i,5] float32 arrays.  Returns (losses dict, aux dict)."""
    B = len(points_list)
    rng = cfg["point_cloud_range"]
    pts, coors = voxelize_batch(points_list, cfg["voxel_size"], rng)
    _, coors_low = voxelize_batch(points_list, cfg["sub_voxel_size_low"], rng)
    _, coors_med = voxelize_batch(points_list, cfg["sub_voxel_size_med"], rng)
    pts_t = torch.as_tensor(pts)
    unq = unique_rows(coors)
    vf, vcoors, _ = vfe_forward(params, pts_t, coors, cfg["voxel_size"], rng, eps=cfg.get("bn_eps", 1e-3), unq=unq)
    if ids_keep is None:
        ids_keep, ids_mask = vanilla_mask_index(vcoors, B, cfg["random_mask_ratio"], generator)
    with torch.no_grad():
        tg = geometric_targets(pts_t, coors, coors_med, coors_low, vcoors, cfg, B, canonical=canonical)
    ik, im = torch.as_tensor(ids_keep), torch.as_tensor(ids_mask)
    tgt = dict(centroid_low=tg["centroid_low"][im], mask_low=tg["mask_low"][im],
               centroid_med=tg["centroid_med"][im], mask_med=tg["mask_med"][im],
               centroid_top=tg["centroid_top"][im], normal=tg["normal"][im])
    pred = backbone_forward(params, vf[ik], vcoors[ids_keep], vcoors[ids_mask], cfg["backbone"])
    losses = forward_loss(pred, tgt, cfg)
    return losses, dict(voxel_feats=vf, voxel_coors=vcoors, targets=tg, pred=pred,
                        ids_keep=ids_keep, ids_mask=ids_mask, coors=coors, coors_med=coors_med,
                        coors_low=coors_low)


# --------------------------------------------------------------------------------------
# configuration constants of configs/mae_sst/m_sst_nus_singlestage_curv_07_..._6x_1e-5.py
# --------------------------------------------------------------------------------------


def mae_sst_cfg(encoder_num_blocks=6, decoder_num_blocks=2, voxel=(0.256, 0.256, 8),
                low=(0.064, 0.064, 1), med=(0.128, 0.128, 2),
                pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), grid=(1, 400, 400)):
    drop_info = {0: dict(max_tokens=56, drop_range=(0, 56)), 1: dict(max_tokens=144, drop_range=(56, 100000))}
    bb = dict(window_shape=(12, 12), shifts_list=[(0, 0), (6, 6)], point_cloud_range=list(pc_range),
              voxel_size=tuple(voxel), d_model=128, nhead=8, dim_feedforward=256,
              encoder_num_blocks=encoder_num_blocks, decoder_num_blocks=decoder_num_blocks,
              drop_info=drop_info, pos_temperature=10000,
              per_sub_voxel_num_low=128, per_sub_voxel_num_med=16)
    return dict(voxel_size=tuple(voxel), sub_voxel_size_low=tuple(low), sub_voxel_size_med=tuple(med),
                point_cloud_range=list(pc_range), grid_size=tuple(grid), spatial_shape=list(grid),
                sub_voxel_ratio_low=(8, 4, 4), sub_voxel_ratio_med=(4, 2, 2), random_mask_ratio=0.7,
                loss_ratio_low=10.0, loss_ratio_med=8.0, loss_ratio_top=10.0, loss_ratio_low_nor=4.0,
                cls_loss_ratio_low=5.0, cls_loss_ratio_med=2.0, bn_eps=1e-3, backbone=bb)


# --------------------------------------------------------------------------------------
# seeded parameters (numpy PCG64 -> identical on every box; names = the reference's
# state_dict keys so one dict feeds the reference, this oracle and the HIP path)
# --------------------------------------------------------------------------------------


def _layer_names(prefix):
    return [(prefix + "win_attn.self_attn.in_proj_weight", (384, 128)),
            (prefix + "win_attn.self_attn.in_proj_bias", (384,)),
            (prefix + "win_attn.self_attn.out_proj.weight", (128, 128)),
            (prefix + "win_attn.self_attn.out_proj.bias", (128,)),
            (prefix + "linear1.weight", (256, 128)), (prefix + "linear1.bias", (256,)),
            (prefix + "linear2.weight", (128, 256)), (prefix + "linear2.bias", (128,)),
            (prefix + "norm1.weight", (128,)), (prefix + "norm1.bias", (128,)),
            (prefix + "norm2.weight", (128,)), (prefix + "norm2.bias", (128,))]


def param_shapes(encoder_num_blocks=6, decoder_num_blocks=2):
    """(name, shape) in the reference's registration order (bb.py:96-130, voxel_encoder.py:161-174)."""
    out = [("voxel_encoder.vfe_layers.0.norm.weight", (64,)), ("voxel_encoder.vfe_layers.0.norm.bias", (64,)),
           ("voxel_encoder.vfe_layers.0.linear.weight", (64, 11)),
           ("voxel_encoder.vfe_layers.1.norm.weight", (128,)), ("voxel_encoder.vfe_layers.1.norm.bias", (128,)),
           ("voxel_encoder.vfe_layers.1.linear.weight", (128, 128)),
           ("backbone.mask_token", (1, 128))]
    for stack, n in (("encoder_blocks", encoder_num_blocks), ("decoder_centroid_blocks", decoder_num_blocks),
                     ("decoder_density_blocks", decoder_num_blocks)):
        for i in range(n):
            for j in range(2):
                out += _layer_names(f"backbone.{stack}.{i}.encoder_list.{j}.")
    for name, o in (("decoder_pred_low", 384), ("decoder_pred_med", 48), ("decoder_pred_top", 3),
                    ("decoder_pred_density_top", 3), ("cls_pred_low", 256), ("cls_pred_med", 32)):
        out += [(f"backbone.{name}.weight", (o, 128)), (f"backbone.{name}.bias", (o,))]
    return out


def make_params(seed, encoder_num_blocks=6, decoder_num_blocks=2, perturb=True):
    """Reference-style init (xavier_uniform on every backbone matrix incl. mask_token, bb.py:318-321;
    torch-default Linear/BN init elsewhere).  perturb=True additionally randomises biases and
    norm affines so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes(encoder_num_blocks, decoder_num_blocks):
        if len(shape) == 2:
            fan_out, fan_in = shape
            if name.startswith("backbone."):
                a = math.sqrt(6.0 / (fan_in + fan_out))
            else:
                a = 1.0 / math.sqrt(fan_in)
            v = rng.uniform(-a, a, shape)
        elif name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            v = np.ones(shape) + (0.1 * rng.standard_normal(shape) if perturb else 0.0)
        else:
            v = 0.05 * rng.standard_normal(shape) if perturb else np.zeros(shape)
        p[name] = torch.tensor(np.asarray(v, np.float32))
    return p


# ---------------------------------------------------------------------------------- N3 DynamicScatter
def dynamic_point_to_voxel(feats, coors, reduce_type="max"):
    """CPU restatement of dynamic_point_to_voxel_forward_gpu (ops/voxel/src/scatter_points_cuda.cu:183-241):
    rows with any negative coordinate are dropped (:199 masked_fill), the rest grouped by at::unique_dim(sorted)
    (:201-202); features reduced by max / sum / mean (:75-97, :233).  -> reduced [M,C], out_coors [M,ndim],
    coors_map [N] (-1 = dropped), reduce_count [M].  Same brute force as the reference's own test
    (tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:56-64)."""
    feats = np.asarray(feats, dtype=np.float32)
    coors = np.asarray(coors, dtype=np.int64)
    n, C = feats.shape
    valid = (coors >= 0).all(axis=1)
    cmap = np.full(n, -1, dtype=np.int32)
    if not valid.any():
        return (np.zeros((0, C), np.float32), np.zeros((0, coors.shape[1]), np.int32), cmap, np.zeros(0, np.int32))
    uq, inv, cnt = np.unique(coors[valid], axis=0, return_inverse=True, return_counts=True)
    cmap[valid] = inv.reshape(-1).astype(np.int32)
    M = uq.shape[0]
    red = np.zeros((M, C), dtype=np.float64)
    if reduce_type == "max":
        red[:] = -np.inf
        np.maximum.at(red, cmap[valid], feats[valid].astype(np.float64))
    else:
        np.add.at(red, cmap[valid], feats[valid].astype(np.float64))
        if reduce_type == "mean":
            red /= cnt[:, None]
    return red.astype(np.float32), uq.astype(np.int32), cmap, cnt.astype(np.int32)


def dynamic_point_to_voxel_grad(grad_reduced, feats, reduced, cmap, cnt, reduce_type="max"):
    """dynamic_point_to_voxel_backward_gpu (scatter_points_cuda.cu:243-310): sum copies, mean divides by the count
    (:99-133), max sends the gradient of (voxel, channel) to the LOWEST-index point equal to the maximum (:135-179)."""
    feats = np.asarray(feats, dtype=np.float32)
    n, C = feats.shape
    g = np.zeros((n, C), dtype=np.float32)
    ok = cmap >= 0
    if reduce_type in ("sum", "mean"):
        g[ok] = grad_reduced[cmap[ok]]
        if reduce_type == "mean":
            g[ok] /= cnt[cmap[ok]][:, None].astype(np.float32)
        return g
    M = reduced.shape[0]
    src = np.full((M, C), n, dtype=np.int64)
    for i in np.nonzero(ok)[0]:
        eq = feats[i] == reduced[cmap[i]]
        src[cmap[i]][eq] = np.minimum(src[cmap[i]][eq], i)
    for m in range(M):
        for c in range(C):
            if src[m, c] < n:
                g[src[m, c], c] = grad_reduced[m, c]
    return g


# ---------------------------------------------------------------------------------- N3 hard voxelization
def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """CPU restatement of hard_voxelize_cpu / hard_voxelize_kernel (ops/voxel/src/voxelization_cpu.cpp:42-132):
    grid = round((max - min) / vs) in fp32 (:113-116); coordinates by the dynamic kernel with THAT grid (clamped,
    :19-36); points walked in index order, a voxel is created on first sight of its cell while fewer than max_voxels
    exist (:66-79), each voxel keeps its first max_points points (:82-88).
    -> voxels [M, max_points, C] f32 zero padded, coors [M,3] int32 (z,y,x), num_points_per_voxel [M] int32."""
    pts = np.asarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    rng = np.asarray(coors_range, dtype=np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    c = np.floor((pts[:, :3] - rng[:3]) / vs).astype(np.int64)
    c = np.clip(c, 0, grid - 1)[:, ::-1]                       # (z, y, x)
    lut, coors, counts = {}, [], []
    voxels = np.zeros((max_voxels, max_points, pts.shape[1]), dtype=np.float32)
    for i in range(pts.shape[0]):
        key = (int(c[i, 0]), int(c[i, 1]), int(c[i, 2]))
        v = lut.get(key, -1)
        if v == -1:
            if len(coors) >= max_voxels:
                continue
            v = len(coors)
            lut[key] = v
            coors.append(key)
            counts.append(0)
        if counts[v] < max_points:
            voxels[v, counts[v]] = pts[i]
            counts[v] += 1
    M = len(coors)
    return voxels[:M], np.asarray(coors, dtype=np.int32).reshape(M, 3), np.asarray(counts, dtype=np.int32)


# ---------------------------------------------------------------------------------- N2 input pipeline
def train_pipeline_cpu(frame, draw, point_cloud_range, sweeps_num=9, remove_close=True, pad_empty_sweeps=True,
                       close_radius=1.0):
    """CPU restatement of the per-sample train pipeline with the random decisions given (`draw`: sweep_choices,
    rotation, scale, translation, flip_horizontal, flip_vertical): LoadPointsFromMultiSweeps.__call__
    (datasets/pipelines/loading.py:184-233; numpy, fp64 products stored into the fp32 array), GlobalRotScaleTrans
    (transforms_3d.py:734-757 via core/points/base_points.py:139-179,186-205,263-269; torch fp32), RandomFlip3D
    (lidar_points.py:28-33), PointsRangeFilter (base_points.py:207-229).  PointShuffle is a permutation: the caller
    compares as a multiset.  -> [N,5] fp32 in concatenation order."""
    key = np.array(frame["points"], dtype=np.float32, copy=True)
    key[:, 4] = 0

    def not_close(p):
        return p[~((np.abs(p[:, 0]) < close_radius) & (np.abs(p[:, 1]) < close_radius))]
    parts = [key]
    sweeps = frame.get("sweeps", [])
    if pad_empty_sweeps and len(sweeps) == 0:
        for _ in range(sweeps_num):
            parts.append(not_close(key) if remove_close else key)
    else:
        ts = frame["timestamp"]
        for idx in draw.sweep_choices:
            sw = sweeps[idx]
            p = np.copy(np.asarray(sw["points"], dtype=np.float32)).reshape(-1, key.shape[1])
            if remove_close:
                p = not_close(p)
            p[:, :3] = p[:, :3] @ np.asarray(sw["sensor2lidar_rotation"], dtype=np.float64).T
            p[:, :3] += np.asarray(sw["sensor2lidar_translation"], dtype=np.float64)
            p[:, 4] = ts - sw["timestamp"] / 1e6
            parts.append(p)
    t = torch.from_numpy(np.concatenate(parts, axis=0))
    rot = t.new_tensor(draw.rotation)
    s, c = torch.sin(rot), torch.cos(rot)
    rot_mat_T = rot.new_tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]]).T
    t[:, :3] = t[:, :3] @ rot_mat_T
    t[:, :3] *= draw.scale
    t[:, :3] += t.new_tensor(draw.translation)
    if draw.flip_horizontal:
        t[:, 1] = -t[:, 1]
    if draw.flip_vertical:
        t[:, 0] = -t[:, 0]
    r = np.asarray(point_cloud_range, dtype=np.float32)
    m = (t[:, 0] > r[0]) & (t[:, 1] > r[1]) & (t[:, 2] > r[2]) & (t[:, 0] < r[3]) & (t[:, 1] < r[4]) & (t[:, 2] < r[5])
    return t[m].numpy()


# ---------------------------------------------------------------------------------- N1 fine-tune fixtures
def seeded_state(seed, shapes):
    """Deterministic values for a state_dict given {name: shape}: the golden generator (reference modules) and the
    tests (this package's modules) call it with the same names, so both load identical weights."""
    import zlib
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if name.endswith("num_batches_tracked"):
            v = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_var"):
            v = (1.0 + 0.1 * np.abs(rs.standard_normal(shape))).astype(np.float32)
        elif len(shape) <= 1 and ("norm" in name or name.split(".")[-2].isdigit()) and name.endswith("weight"):
            v = (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif len(shape) <= 1:
            v = (0.02 * rs.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = (rs.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        out[name] = torch.from_numpy(v)
    return out
