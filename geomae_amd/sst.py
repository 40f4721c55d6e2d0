"""SST blocks and the MAE backbone of the mae_sst config.

Reference: mmdet3d/models/sst/sst_basic_block.py:13-147 (WindowAttention / EncoderLayer /
BasicShiftBlock) and mmdet3d/models/backbones/multi_mae_sst_spearate_top_only.py (registered as
MultiMAESSTSPChoose).  Parameter names match the reference's state_dict
(backbone.encoder_blocks.{i}.encoder_list.{j}.win_attn.self_attn.in_proj_weight, ...), so
checkpoints interchange with the fine-tune configs.

Differences in mechanism, not in math: windows are CSR segments built once per token set by
libgeomae_hip (no zero-padded [W, 56|144, C] buckets, no .item() syncs, no per-layer
flat2window/window2flat copies); the positional embedding is a 144 x 128 table gathered per token
(it is a pure function of the in-window coordinate, bb.py:361-394); attention inside a window is
the hand-written MFMA kernel (ops.window_attention).  The dense projections / FFN run as bf16 MFMA
GEMMs with fp32 accumulation and an fp32 residual stream (compute_dtype='bf16', BASELINE config 2) or
in fp32 (compute_dtype='fp32', used by the tight parity tests).
"""
import math

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .registry import BACKBONES


def pos_embed_table(window_shape, d_model, temperature=10000):
    """[wx*wy, d_model] fp32, row cx*wy + cy  (bb.py:361-394)."""
    wx, wy = window_shape
    cx = torch.arange(wx, dtype=torch.float32).repeat_interleave(wy)
    cy = torch.arange(wy, dtype=torch.float32).repeat(wx)
    x, y = cx - wx / 2, cy - wy / 2
    pos_length = d_model // 2
    inv_freq = torch.arange(pos_length, dtype=torch.float32)
    inv_freq = temperature ** (2 * (inv_freq // 2) / pos_length)
    ex = x[:, None] / inv_freq[None, :]
    ey = y[:, None] / inv_freq[None, :]
    ex = torch.stack([ex[:, ::2].sin(), ex[:, 1::2].cos()], dim=-1).flatten(1)
    ey = torch.stack([ey[:, ::2].sin(), ey[:, 1::2].cos()], dim=-1).flatten(1)
    return torch.cat([ex, ey], dim=-1)


class PackedLayers:
    """bf16 MFMA-layout copies of every SST layer's matrices (geomae_pack_weights), refreshed once per
    forward, plus the per-layer C structs handed to the fused kernels."""
    PER_LAYER = 262144   # bf16 elements: wqkv 49152 | wqkT 32768 | wvT 16384 | wo, woT 16384 x2 | w1, w1T, w2, w2T 32768 x4

    HEAD_ROWS = (("decoder_pred_low", 0), ("cls_pred_low", 384), ("decoder_pred_med", 640), ("cls_pred_med", 688),
                 ("decoder_pred_top", 720), ("decoder_pred_density_top", 768))   # row offsets in the [800,128] block

    def __init__(self, layers, backbone=None):
        self.layers = list(layers)
        self.backbone = backbone          # owner of the six head Linears (None: layers only)
        self.key = None
        self.packed = None

    def _build(self, dev):
        from ._lib import GeomaeSstLayerWeights
        n = len(self.layers)
        # [row-major blocks of the n layers | heads 800 x 128 | fragment-major blocks of the n layers]
        self.packed = torch.zeros(2 * n * self.PER_LAYER + 800 * 128, dtype=torch.bfloat16, device=dev)
        self.head_w = self.packed[n * self.PER_LAYER:n * self.PER_LAYER + 800 * 128]
        frag0 = n * self.PER_LAYER + 800 * 128
        self.head_bias = torch.zeros(800, dtype=torch.float32, device=dev)
        base = self.packed.data_ptr()
        desc, self.structs = [], []

        def f(t):
            return t.data_ptr() // 4

        for i, L in enumerate(self.layers):
            a = L.win_attn.self_attn
            win, wo, w1, w2 = a.in_proj_weight, a.out_proj.weight, L.linear1.weight, L.linear2.weight
            off = i * self.PER_LAYER
            o = dict(wqkv=off, wqkT=off + 49152, wvT=off + 81920, wo=off + 98304, woT=off + 114688, w1=off + 131072,
                     w1T=off + 163840, w2=off + 196608, w2T=off + 229376)
            desc += [[f(win), 384, 128, 0, o["wqkv"]], [f(win), 256, 128, 1, o["wqkT"]],
                     [f(win) + 256 * 128, 128, 128, 1, o["wvT"]], [f(wo), 128, 128, 0, o["wo"]],
                     [f(wo), 128, 128, 1, o["woT"]], [f(w1), 256, 128, 0, o["w1"]], [f(w1), 256, 128, 1, o["w1T"]],
                     [f(w2), 128, 256, 0, o["w2"]], [f(w2), 128, 256, 1, o["w2T"]]]
            # the same nine matrices fragment-major (descriptor mode | 4) for the one-launch layer kernels
            desc += [[d[0], d[1], d[2], d[3] | 4, d[4] - off + frag0 + off] for d in desc[-9:]]
            st = GeomaeSstLayerWeights()
            for k, v in o.items():
                setattr(st, k + "_p", base + 2 * v)
            st.frag_p = base + 2 * (frag0 + off)
            st.bqkv, st.bo = a.in_proj_bias.data_ptr(), a.out_proj.bias.data_ptr()
            st.b1, st.b2 = L.linear1.bias.data_ptr(), L.linear2.bias.data_ptr()
            st.ln1_w, st.ln1_b = L.norm1.weight.data_ptr(), L.norm1.bias.data_ptr()
            st.ln2_w, st.ln2_b = L.norm2.weight.data_ptr(), L.norm2.bias.data_ptr()
            st.d_model, st.d_ffn, st.ln_eps = 128, 256, L.norm1.eps
            self.structs.append(st)
        if self.backbone is not None:
            for name, row0 in self.HEAD_ROWS:
                lin = getattr(self.backbone, name)
                desc.append([f(lin.weight), lin.weight.shape[0], 128, 0, n * self.PER_LAYER + row0 * 128])
                desc.append([f(lin.bias), lin.bias.shape[0], 1, 2, row0])
        self.desc = torch.tensor(desc, dtype=torch.int64, device=dev)
        self.n_desc = len(desc)

    def _key(self):
        heads = () if self.backbone is None else tuple(
            getattr(self.backbone, n).weight.data_ptr() for n, _ in self.HEAD_ROWS)
        return tuple(p.data_ptr() for L in (self.layers[0], self.layers[-1]) for p in L.parameters()) + heads + \
            (self.layers[0].linear1.weight.device,)

    def refresh(self):
        """Re-pack from the current fp32 master weights (they change every optimizer step)."""
        k = self._key()
        if k != self.key:
            self._build(k[-1])
            self.key = k
        ev, self.ready = getattr(self, "ready", None), None
        if ev is not None:                   # an earlier pack-ahead on another stream: do not race it
            torch.cuda.current_stream().wait_event(ev)
        ops.pack_weights(self.desc, self.n_desc, 384 * 128, self.packed, self.head_bias)
        self._prepacked = None

    # ---- packing ahead: the copies depend on the weights only, so the trainer packs them right after the optimizer
    # step (on a side stream) and the next step's critical path does not contain the pack kernel
    def _sources(self):
        srcs = getattr(self, "_srcs", None)
        if srcs is None:                       # the Parameter objects are stable (only their .data / .grad move)
            srcs = [p for L in self.layers for p in L.parameters()]
            if self.backbone is not None:
                for n, _ in self.HEAD_ROWS:
                    lin = getattr(self.backbone, n)
                    srcs += [lin.weight, lin.bias]
            self._srcs = srcs
        return srcs

    def _versions(self):
        return tuple(p._version for p in self._sources())

    def prepack(self):
        """Pack now for the NEXT forward; valid for one `refresh_if_stale` as long as no torch op writes a source
        parameter in between (tensor version counters; writers that bypass them call `invalidate`)."""
        self.refresh()
        self._prepacked = self._versions()

    def invalidate(self):
        self._prepacked = None

    def refresh_if_stale(self):
        """-> event to wait for before reading the packed copies (None: packed on the current stream)."""
        v, self._prepacked = getattr(self, "_prepacked", None), None
        if v is not None and self._key() == self.key and v == self._versions():
            ev, self.ready = getattr(self, "ready", None), None
            return ev
        self.refresh()                                       # (waits for a pending pack-ahead itself)
        return None

    def grads(self, layer_index):
        """C struct of gradient pointers; allocates .grad where autograd has not yet."""
        from ._lib import GeomaeSstLayerGrads
        L = self.layers[layer_index]
        a = L.win_attn.self_attn
        ps = dict(wqkv=a.in_proj_weight, bqkv=a.in_proj_bias, wo=a.out_proj.weight, bo=a.out_proj.bias,
                  w1=L.linear1.weight, b1=L.linear1.bias, w2=L.linear2.weight, b2=L.linear2.bias,
                  ln1_w=L.norm1.weight, ln1_b=L.norm1.bias, ln2_w=L.norm2.weight, ln2_b=L.norm2.bias)
        g = GeomaeSstLayerGrads()
        for k, p in ps.items():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            setattr(g, k, p.grad.data_ptr())
        return g


    def weight_array(self, first, count):
        from ._lib import GeomaeSstLayerWeights
        return (GeomaeSstLayerWeights * count)(*self.structs[first:first + count])

    def grad_array(self, first, count):
        """ctypes array of per-layer gradient-pointer structs; cached while the first and last gradient of the range
        stay where they were (the trainer keeps every .grad as a view of one flat buffer)."""
        from ._lib import GeomaeSstLayerGrads
        cache = self.__dict__.setdefault("_grad_arrays", {})
        a, b = self.layers[first].linear1.weight.grad, self.layers[first + count - 1].norm2.bias.grad
        tag = (a.data_ptr(), b.data_ptr()) if (a is not None and b is not None) else None
        hit = cache.get((first, count))
        if hit is not None and tag is not None and hit[0] == tag:
            return hit[1]
        arr = (GeomaeSstLayerGrads * count)(*[self.grads(i) for i in range(first, first + count)])
        a, b = self.layers[first].linear1.weight.grad, self.layers[first + count - 1].norm2.bias.grad
        cache[(first, count)] = ((a.data_ptr(), b.data_ptr()), arr)
        return arr

    def head_grads(self):
        from ._lib import GeomaeHeadGrads
        g = GeomaeHeadGrads()
        names = dict(reg_low="decoder_pred_low", cls_low="cls_pred_low", reg_med="decoder_pred_med",
                     cls_med="cls_pred_med", reg_top="decoder_pred_top", nor_top="decoder_pred_density_top")
        for k, attr in names.items():
            lin = getattr(self.backbone, attr)
            for suffix, p in (("_w", lin.weight), ("_b", lin.bias)):
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                setattr(g, k + suffix, p.grad.data_ptr())
        return g


class _HeadsLoss(torch.autograd.Function):
    """Six heads + six losses + their backward in one kernel (ops.heads_loss).  Returns the [6] loss vector
    (curv_around, centroid_low, centroid_med, centroid_top, cls_low, cls_med), each already multiplied by its
    loss ratio; gradients assume the objective is their plain sum (as mmdet's _parse_losses forms it)."""

    @staticmethod
    def forward(ctx, cen, den, packed, n_keep, n_mask, tgt, weights):
        losses, d_cen, d_den, saved = ops.heads_loss(cen.contiguous(), den.contiguous(), n_keep, n_mask, packed.head_w,
                                                     packed.head_bias, tgt, weights)
        ctx.packed, ctx.n_mask, ctx.saved = packed, n_mask, saved
        ctx.save_for_backward(d_cen, d_den)
        return losses

    @staticmethod
    def backward(ctx, dl):
        d_cen, d_den = ctx.saved_tensors
        ops.heads_weight_grad(ctx.n_mask, *ctx.saved, ctx.packed.head_grads())
        return d_cen, d_den, None, None, None, None, None


class _FusedStack(torch.autograd.Function):
    """A run of consecutive SST layers (the encoder, or one decoder) as ONE C call forward and ONE backward:
    3 kernels per layer forward (qkv, window attention, out-proj+LN+FFN+LN), 4 backward (ffn backward,
    attention backward, qkv backward, weight gradients).  Parameter gradients are accumulated straight into
    .grad by the kernels (they are not autograd outputs of this Function)."""

    @staticmethod
    def forward(ctx, x, packed, first, count, layouts, pos_table, nhead):
        weights = packed.weight_array(first, count)
        z, saved = ops.sst_stack_forward(x.contiguous(), weights, layouts, pos_table, nhead)
        ctx.packed, ctx.first, ctx.count, ctx.layouts, ctx.pos_table, ctx.nhead = packed, first, count, layouts, pos_table, nhead
        ctx.weights, ctx.n = weights, x.shape[0]
        ctx.save_for_backward(saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        (saved,) = ctx.saved_tensors
        grads = ctx.packed.grad_array(ctx.first, ctx.count)
        dx = ops.sst_stack_backward(dz.contiguous().float(), ctx.n, ctx.weights, grads, ctx.layouts, ctx.pos_table,
                                    ctx.nhead, saved)
        return dx, None, None, None, None, None, None


class _FusedDecoderPair(torch.autograd.Function):
    """The two decoder stacks (centroid / density) read the same tokens and are independent of each other
    (bb.py:269-277): run them concurrently on two HIP streams, forward and backward -- each stack alone is a
    chain of 13-50 us kernels that leaves most CUs idle.  Buffers are allocated on the current stream before
    the fork and both streams are joined before returning, so the caching allocator never sees them."""

    @staticmethod
    def forward(ctx, x, packed, first_a, first_b, count, layouts, pos_table, nhead, streams):
        x = x.contiguous()
        cur = torch.cuda.current_stream()
        wa, wb = packed.weight_array(first_a, count), packed.weight_array(first_b, count)
        for s in streams:
            s.wait_stream(cur)
        za, sa = ops.sst_stack_forward(x, wa, layouts, pos_table, nhead, stream=streams[0])
        zb, sb = ops.sst_stack_forward(x, wb, layouts, pos_table, nhead, stream=streams[1])
        for s in streams:
            cur.wait_stream(s)
        ctx.packed, ctx.firsts, ctx.count, ctx.layouts, ctx.pos_table, ctx.nhead = packed, (first_a, first_b), count, layouts, pos_table, nhead
        ctx.weights, ctx.n, ctx.streams = (wa, wb), x.shape[0], streams
        ctx.save_for_backward(sa, sb)
        return za, zb

    @staticmethod
    def backward(ctx, dza, dzb):
        sa, sb = ctx.saved_tensors
        cur = torch.cuda.current_stream()
        ga, gb = (ctx.packed.grad_array(f, ctx.count) for f in ctx.firsts)
        dza, dzb = dza.contiguous().float(), dzb.contiguous().float()
        for s in ctx.streams:
            s.wait_stream(cur)
        dxa, keep_a = ops.sst_stack_backward(dza, ctx.n, ctx.weights[0], ga, ctx.layouts, ctx.pos_table, ctx.nhead, sa,
                                             stream=ctx.streams[0])
        dxb, keep_b = ops.sst_stack_backward(dzb, ctx.n, ctx.weights[1], gb, ctx.layouts, ctx.pos_table, ctx.nhead, sb,
                                             stream=ctx.streams[1])
        for s in ctx.streams:
            cur.wait_stream(s)
        dx = dxa + dxb
        del keep_a, keep_b          # scratch buffers: released only after both streams were joined
        return dx, None, None, None, None, None, None, None, None


EXACT_ATTENTION_FP32 = True      # compute_dtype='fp32' (the parity mode): attention core in fp32 ATen ops, not the bf16 kernel


def window_attention_fp32(qkv, layout, nhead):
    """softmax(q k^T / sqrt(d)) v inside every window in fp32 ATen ops, differentiable (qkv [n, 3C] fp32 -> [n, C]).
    The parity mode's attention core: with it the composed fp32 path has NO bf16 step left, so the tight-tolerance tests
    separate the error of the hand-written bf16 MFMA kernel (ops.window_attention) from everything else.  Windows are
    padded to the longest one (what the reference does per bucket, sst_basic_block.py:36-59); never on the product path."""
    n, C3 = qkv.shape
    C = C3 // 3
    d = C // nhead
    W = int(layout.num_windows)                                   # (host sync: test path only)
    ws = layout.win_start[:W + 1].long()
    cnt = ws[1:] - ws[:-1]
    T = int(cnt.max())
    w_of_pos = torch.repeat_interleave(torch.arange(W, device=qkv.device), cnt)
    slot = torch.arange(n, device=qkv.device) - ws[w_of_pos]
    tok = layout.win_tokens[:n].long()
    pad = qkv.new_zeros((W, T, C3))
    pad[w_of_pos, slot] = qkv[tok]
    q, k, v = (pad[..., i * C:(i + 1) * C].reshape(W, T, nhead, d).permute(0, 2, 1, 3) for i in range(3))
    valid = torch.arange(T, device=qkv.device)[None, :] < cnt[:, None]                     # [W, T]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(W, T, C)
    out = qkv.new_zeros((n, C))
    out[tok] = o[w_of_pos, slot]
    return out


class WindowAttention(nn.Module):
    def __init__(self, d_model, nhead, dropout, batch_first=False, layer_id=None):
        super().__init__()
        assert dropout == 0.0, "the mae_sst config trains with dropout 0"
        self.nhead = nhead
        self.d_model = d_model
        # parameter container only: in_proj_weight [3C, C], in_proj_bias, out_proj.{weight,bias}
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.layer_id = layer_id

    def forward(self, x, pos, layout, compute_dtype):
        C = self.d_model
        a = self.self_attn
        dt = compute_dtype
        w, b = a.in_proj_weight.to(dt), a.in_proj_bias.to(dt)
        qk_in = (x + pos).to(dt)
        qk = F.linear(qk_in, w[:2 * C], b[:2 * C])
        v = F.linear(x.to(dt), w[2 * C:], b[2 * C:])
        qkv = torch.cat([qk, v], dim=1)
        if dt == torch.float32 and EXACT_ATTENTION_FP32:
            o = window_attention_fp32(qkv, layout, self.nhead)
        else:
            o = ops.window_attention(qkv.to(torch.bfloat16), layout, self.nhead).to(dt)
        return F.linear(o, a.out_proj.weight.to(dt), a.out_proj.bias.to(dt)).float()


class EncoderLayer(nn.Module):
    """Post-norm transformer layer (sst_basic_block.py:63-102)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", batch_first=False,
                 layer_id=None, mlp_dropout=0):
        super().__init__()
        assert not batch_first
        assert activation == "gelu", "the mae_sst config uses GELU"
        self.win_attn = WindowAttention(d_model, nhead, dropout, layer_id=layer_id)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, layout, compute_dtype):
        """Composed (ATen/rocBLAS + window-attention kernel) form of the layer: kept as the A/B debugging
        path for the fused kernels (`compute_dtype='fp32'`); the product path is _FusedLayer."""
        dt = compute_dtype
        src2 = self.win_attn(src, pos, layout, dt)
        src = self.norm1(src + src2)
        h = F.gelu(F.linear(src.to(dt), self.linear1.weight.to(dt), self.linear1.bias.to(dt)))
        src2 = F.linear(h, self.linear2.weight.to(dt), self.linear2.bias.to(dt)).float()
        return self.norm2(src + src2)


class BasicShiftBlock(nn.Module):
    """Two encoder layers, the second on the shifted window layout (sst_basic_block.py:104-147)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", batch_first=False,
                 block_id=-100):
        super().__init__()
        self.encoder_list = nn.ModuleList([
            EncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, batch_first, layer_id=block_id * 2 + i)
            for i in range(2)])

    def forward(self, src, pos_list, layout_list, compute_dtype):
        num_shifts = len(layout_list)
        assert num_shifts in (1, 2)
        out = src
        for i in range(2):
            s = i % num_shifts
            out = self.encoder_list[i](out, pos_list[s], layout_list[s], compute_dtype)
        return out


@BACKBONES.register_module()
class MultiMAESSTSPChoose(nn.Module):
    """MAE-style SST: 6 encoder blocks on visible pillars, mask-token insertion, two 2-block decoder
    stacks (centroid / density) on all pillars, six linear heads on the masked rows
    (multi_mae_sst_spearate_top_only.py:20-303)."""

    def __init__(self, window_shape, shifts_list, point_cloud_range, voxel_size, shuffle_voxels=False, d_model=[],
                 nhead=[], sub_voxel_ratio_low=[], sub_voxel_ratio_med=[], cls_sub_voxel=False,
                 encoder_num_blocks=6, decoder_num_blocks=2, dim_feedforward=[], dropout=0.0, activation="gelu",
                 output_shape=None, low=True, med=True, top=True, debug=True, drop_info=None, normalize_pos=False,
                 pos_temperature=10000, in_channel=None, conv_kwargs=None, checkpoint_blocks=[],
                 compute_dtype="bf16"):
        super().__init__()
        assert drop_info is not None
        assert not shuffle_voxels and not normalize_pos and in_channel is None
        self.shifts_list = shifts_list
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.meta_drop_info = drop_info
        self.pos_temperature = pos_temperature
        self.d_model = d_model
        self.window_shape = tuple(window_shape)
        self.nhead = nhead
        self.cls_sub_voxel = cls_sub_voxel
        self.low, self.med, self.top = low, med, top
        self.debug = debug
        self.output_shape = output_shape
        self.compute_dtype = compute_dtype
        assert len(set(d_model)) == 1

        def blocks(n):
            return nn.ModuleList([BasicShiftBlock(d_model[i], nhead[i], dim_feedforward[i], dropout, activation,
                                                  batch_first=False, block_id=i) for i in range(n)])
        self.encoder_blocks = blocks(encoder_num_blocks)
        self.decoder_centroid_blocks = blocks(decoder_num_blocks)
        self.decoder_density_blocks = blocks(decoder_num_blocks)
        self.mask_token = nn.Parameter(torch.zeros(1, d_model[-1]))
        self.per_sub_voxel_num_low = int(np.prod(sub_voxel_ratio_low))
        self.per_sub_voxel_num_med = int(np.prod(sub_voxel_ratio_med))
        D = d_model[-1]
        self.decoder_pred_low = nn.Linear(D, self.per_sub_voxel_num_low * 3)
        self.decoder_pred_med = nn.Linear(D, self.per_sub_voxel_num_med * 3)
        self.decoder_pred_top = nn.Linear(D, 3)
        if low:
            self.decoder_pred_density_low = nn.Linear(D, self.per_sub_voxel_num_low * 3)
        if med:
            self.decoder_pred_density_med = nn.Linear(D, self.per_sub_voxel_num_med * 3)
        if top:
            self.decoder_pred_density_top = nn.Linear(D, 3)
        if cls_sub_voxel:
            self.cls_pred_low = nn.Linear(D, self.per_sub_voxel_num_low * 2)
            self.cls_pred_med = nn.Linear(D, self.per_sub_voxel_num_med * 2)
        self._reset_parameters()
        self.register_buffer("pos_table", pos_embed_table(self.window_shape, d_model[0], pos_temperature),
                             persistent=False)
        bev_x = int(np.ceil((point_cloud_range[3] - point_cloud_range[0]) / voxel_size[0]))
        bev_y = int(np.ceil((point_cloud_range[4] - point_cloud_range[1]) / voxel_size[1]))
        shift = shifts_list[1] if len(shifts_list) > 1 else (0, 0)
        self._wcfg = ops.make_window_config(self.window_shape, shift, (bev_x, bev_y))
        # every window must fit its bucket, i.e. no token is ever dropped in training:
        # the largest training bucket must hold a full window
        all_layers = [l for stack in (self.encoder_blocks, self.decoder_centroid_blocks, self.decoder_density_blocks)
                      for b in stack for l in b.encoder_list]
        self._packed = PackedLayers(all_layers, backbone=self)
        self.concurrent_decoders = True
        self._streams = None
        self._stack_base = {"enc": 0, "cen": 2 * encoder_num_blocks,
                            "den": 2 * (encoder_num_blocks + decoder_num_blocks)}
        info = drop_info[0] if isinstance(drop_info, tuple) else drop_info
        assert max(v["max_tokens"] for v in info.values()) >= self.window_shape[0] * self.window_shape[1], \
            "token dropping is not supported: the largest bucket must hold a full window"

    def _reset_parameters(self):
        for name, p in self.named_parameters():
            if p.dim() > 1 and "scaler" not in name:
                nn.init.xavier_uniform_(p)

    def _dtype(self):
        return torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32

    @property
    def fused(self):
        return self.compute_dtype == "bf16"

    def get_voxel_info(self, coors, batch_size):
        """CSR window layouts for both shifts (+ gathered positional embeddings for the composed path)."""
        coors = coors.int().contiguous()
        ns = len(self.shifts_list)
        if ns <= 4:          # one launch per build stage, and the attention plan the one-launch layer kernel needs
            layouts = ops.window_build_batch([(coors, s) for s in range(ns)], batch_size, self._wcfg)
        else:
            layouts = [ops.window_build(coors, batch_size, self._wcfg, s) for s in range(ns)]
        pos = None if self.fused else [self.pos_table[L.tok_pos[:L.n].long()] for L in layouts]
        return layouts, pos

    def _run_stack(self, blocks, name, x, pos, layouts):
        if self.fused:
            return _FusedStack.apply(x, self._packed, self._stack_base[name], 2 * len(blocks), layouts, self.pos_table,
                                     self.nhead[0])
        dt = self._dtype()
        for block in blocks:
            x = block(x, pos, layouts, dt)
        return x

    def forward(self, voxel_feat, coors, coors_mask, batch_size):
        if self.fused:
            self._packed.refresh()
        layouts, pos = self.get_voxel_info(coors, batch_size)
        x = self.forward_encoder(voxel_feat.float(), layouts, pos)
        return self.forward_decoder(x, coors, coors_mask, batch_size)

    def forward_encoder(self, x, layouts, pos):
        return self._run_stack(self.encoder_blocks, "enc", x, pos, layouts)

    def build_layouts(self, coors, coors_mask, batch_size, coors_all=None):
        """Window layouts of the encoder tokens (kept pillars) and of the decoder tokens (kept + masked): they depend
        on coordinates only, so the detector builds them on a side stream under the VFE forward.
        coors_all: optional [n_keep + n_mask, 4] int32 = cat(coors, coors_mask) already in one buffer (then `coors` /
        `coors_mask` may be None and `coors_all[:n_keep]` is passed as coors)."""
        ns = len(self.shifts_list)
        if coors_all is not None and self.fused and 2 * ns <= 4:
            c_enc = coors if coors is not None else coors_all
            L = ops.window_build_batch([(c_enc, s) for s in range(ns)] + [(coors_all, s) for s in range(ns)], batch_size,
                                       self._wcfg)
            return L[:ns], L[ns:]
        if not self.fused or 2 * ns > 4:
            enc, _ = self.get_voxel_info(coors, batch_size)
            dec, _ = self.get_voxel_info(torch.cat([coors, coors_mask], dim=0), batch_size)
            return enc, dec
        c_enc = coors.int().contiguous()
        c_dec = torch.cat([coors, coors_mask], dim=0).int().contiguous()
        L = ops.window_build_batch([(c_enc, s) for s in range(ns)] + [(c_dec, s) for s in range(ns)], batch_size, self._wcfg)
        return L[:ns], L[ns:]

    def forward_losses(self, voxel_feat, coors, coors_mask, batch_size, tgt, loss_weights, layouts=None):
        """Fused training path: encoder, both decoder stacks, then heads + losses (+ their backward) in one
        kernel -- the eight prediction tensors of forward() are never materialised.  Returns the [6] losses."""
        assert self.fused and self.cls_sub_voxel and self.top and not self.low and not self.med
        self._packed.refresh()
        enc_layouts, dec_layouts = layouts if layouts is not None else self.build_layouts(coors, coors_mask, batch_size)
        x = self.forward_encoder(voxel_feat.float(), enc_layouts, None)
        cen, den = self.decode(x, coors, coors_mask, batch_size, layouts=dec_layouts)
        return _HeadsLoss.apply(cen, den, self._packed, coors.shape[0], coors_mask.shape[0], tgt, loss_weights)

    # ---- explicit (autograd-free) schedule: the forward keeps what the backward needs, the backward is called by
    # hand in reverse order.  Same kernels and the same parameter-gradient accumulation as the autograd Functions
    # above; no tape, no engine thread, no gradient seeds / index_add / cat nodes.
    @torch.no_grad()
    def losses_and_grads_explicit(self, voxel_feat, n_mask, batch_size, tgt, loss_weights, layouts, on_early_grads=None,
                                  packed_fresh=False, tgt_ready=None, bufs=None, keep_rows=None, on_encoder_grads=None):
        """-> ([6] losses, d_voxel_feat [n_keep,128]); parameter gradients are accumulated into .grad.
        keep_rows (int32 ids_keep): voxel_feat is then ALL pillars' features [V,128] and the encoder gathers / its
        backward scatters the kept rows itself (bb.py:178); the gradient comes back as [V,128] in bufs["d_vf"]
        (zeroed by the caller: masked pillars get no gradient).
        on_early_grads(): called once the gradients of the heads, both decoders and the mask token are enqueued
        (everything except the encoder), so that the caller can start exchanging them; on_encoder_grads(): likewise
        once the encoder's are.  Both are called with the stream current behind which those gradients are complete.
        packed_fresh: the caller already re-packed the bf16 weights for this step (on another stream, ordered before
        this call's stream).  tgt_ready: event after which `tgt` may be read (targets built on a side stream).
        bufs: optional dict prepared off the critical path by the caller (detector.train_step_explicit): zeroed
        "d_cen" / "d_den" [n_keep + n_mask, 128], "losses" [6], optionally "ready" (event after which they may be used; omitted when the caller's stream is already
        ordered behind their preparation) and "side" (the stream for work nobody waits on
        until the optimizer: the mask-token gradient reduction)."""
        assert self.fused and self.cls_sub_voxel and self.top and not self.low and not self.med
        P, nh, pt = self._packed, self.nhead[0], self.pos_table
        if not packed_fresh:
            P.refresh()
        enc_layouts, dec_layouts = layouts
        n_keep = voxel_feat.shape[0] if keep_rows is None else keep_rows.numel()
        scatter = None if keep_rows is None else (keep_rows, bufs["d_vf"])
        n_enc, n_dec = 2 * len(self.encoder_blocks), 2 * len(self.decoder_centroid_blocks)
        w_enc = P.weight_array(self._stack_base["enc"], n_enc)
        w_cen, w_den = P.weight_array(self._stack_base["cen"], n_dec), P.weight_array(self._stack_base["den"], n_dec)
        cur = torch.cuda.current_stream()
        # the decoders' input = encoder output followed by n_mask copies of the mask token: the stacks take that as
        # (rows, tail) and never materialise the concatenation (bb.py:239-246)
        z_enc, s_enc = ops.sst_stack_forward(voxel_feat.float().contiguous(), w_enc, enc_layouts, pt, nh, rows=keep_rows)
        dec_tail = (self.mask_token.detach(), n_mask)
        if bufs is None:
            d_out = losses_buf = None
        else:
            d_out, losses_buf = (bufs["d_cen"], bufs["d_den"]), bufs.get("losses")
            if bufs.get("d_cen2") is not None:              # the split form of the heads kernel (two summands of d_cen)
                d_out = (bufs["d_cen"], bufs["d_cen2"], bufs["d_den"])
            if bufs.get("ready") is not None:
                cur.wait_event(bufs["ready"])
        ops.mark("enc_fwd_done")
        if self._streams is None:
            self._streams = (None, ops.side_streams()["dec_b"])
        # The two decoder stacks run concurrently: one on a side stream, the other on the current stream itself.  A
        # cross-queue wait costs ~10 us of queue time even when its event has long fired (tools/archive/phase_events.py), so
        # the fork / join is one wait on each side instead of two.
        _, sb_ = self._streams
        sb_.wait_stream(cur)
        den, s_den = ops.sst_stack_forward(z_enc, w_den, dec_layouts, pt, nh, stream=sb_, tail=dec_tail)
        cen, s_cen = ops.sst_stack_forward(z_enc, w_cen, dec_layouts, pt, nh, tail=dec_tail)
        cur.wait_stream(sb_)
        if tgt_ready is not None:
            cur.wait_event(tgt_ready)
        ops.mark("dec_fwd_done")
        split = d_out is not None and len(d_out) == 3
        losses, d_cen, d_den, saved_h = ops.heads_loss(cen, den, n_keep, n_mask, P.head_w, P.head_bias, tgt, loss_weights,
                                                       d_out=d_out, losses=losses_buf, split=split)
        d_cen, d_cen2 = d_cen if split else (d_cen, None)
        # ---------------- backward
        side = bufs.get("side") if bufs is not None else None
        if side is None:
            ops.heads_weight_grad(n_mask, *saved_h, P.head_grads())
        else:
            # the heads' weight-gradient contraction feeds nothing but the optimizer: off the main stream, beside the
            # decoder backward (the stream is joined before the optimizer for the mask-token gradient anyway)
            side.wait_stream(cur)
            for t in saved_h:
                t.record_stream(side)
            with torch.cuda.stream(side):
                ops.heads_weight_grad(n_mask, *saved_h, P.head_grads())
        g_cen, g_den = P.grad_array(self._stack_base["cen"], n_dec), P.grad_array(self._stack_base["den"], n_dec)
        n = n_keep + n_mask
        ops.mark("heads_done")
        sb_.wait_stream(cur)
        # the masked rows of both decoders' input gradients sum into the mask-token gradient: each stack's last data
        # kernel adds its rows' column sums (tail_sum), nobody materialises or reduces dxa + dxb
        if self.mask_token.grad is None:
            self.mask_token.grad = torch.zeros_like(self.mask_token)
        mt = (self.mask_token.grad, n_keep)
        if side is None:
            dxb, keep_b = ops.sst_stack_backward(d_den, n, w_den, g_den, dec_layouts, pt, nh, s_den, stream=sb_, tail_sum=mt)
            dxa = ops.sst_stack_backward(d_cen, n, w_cen, g_cen, dec_layouts, pt, nh, s_cen, tail_sum=mt, dz_add=d_cen2)
        else:
            # each stack's last kernel, the first layer's weight-gradient contraction (~50 us at decoder size, read
            # only by the optimizer), runs on the side stream instead of closing the decoder backward: it overlaps
            # the encoder backward, whose launches leave most CUs idle
            dxb, keep_b = ops.sst_stack_backward(d_den, n, w_den, g_den, dec_layouts, pt, nh, s_den, stream=sb_, defer_last=True,
                                                 tail_sum=mt)
            side.wait_stream(sb_)
            with torch.cuda.stream(side):
                ops.flush_weight_grad()
            dxa, keep_a = ops.sst_stack_backward(d_cen, n, w_cen, g_cen, dec_layouts, pt, nh, s_cen, defer_last=True,
                                                 tail_sum=mt, dz_add=d_cen2)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.flush_weight_grad()
            for t in (keep_a, keep_b, s_cen, s_den):
                t.record_stream(side)
            del keep_a
        cur.wait_stream(sb_)
        ops.mark("dec_bwd_done")
        # the token gradient is dxa + dxb: its kept rows are summed by the encoder backward's first kernel (dz_add)
        del keep_b
        if on_early_grads is not None:
            if side is None:
                on_early_grads()
            else:
                # every gradient of the heads, the decoders and the mask token is complete in the geometry stream's
                # order (it waited for both decoder backwards, whose last kernels also add up the mask-token gradient,
                # before the two flushes, and carries the heads' contraction itself): the collective is ordered behind
                # THAT stream, the main stream never waits for it
                with torch.cuda.stream(side):
                    on_early_grads()
        g_enc = P.grad_array(self._stack_base["enc"], n_enc)
        if side is None:
            d_vf = ops.sst_stack_backward(dxa, n_keep, w_enc, g_enc, enc_layouts, pt, nh, s_enc, scatter=scatter,
                                          dz_add=dxb)
        else:
            # the first layer's weight-gradient contraction (the stack's last kernel) goes to the side stream, beside
            # the VFE backward; the caller joins `side` before the optimizer (bufs["join_side"])
            d_vf, keep = ops.sst_stack_backward(dxa, n_keep, w_enc, g_enc, enc_layouts, pt, nh, s_enc, defer_last=True,
                                                scatter=scatter, dz_add=dxb)
            side.wait_stream(cur)
            keep.record_stream(side)
            s_enc.record_stream(side)
            with torch.cuda.stream(side):
                ops.flush_weight_grad()
                if on_encoder_grads is not None:          # (the flush was the encoder's last gradient kernel)
                    on_encoder_grads()
            bufs["join_side"] = True
        if side is None and on_encoder_grads is not None:
            on_encoder_grads()
        ops.mark("enc_bwd_done")
        return losses, d_vf

    def decode(self, visible_voxel_feat, coors, coors_mask, batch_size, layouts=None):
        mask_tokens = self.mask_token.repeat(coors_mask.shape[0], 1)
        tokens = torch.cat([visible_voxel_feat, mask_tokens], dim=0)
        pos = None
        if layouts is None:
            layouts, pos = self.get_voxel_info(torch.cat([coors, coors_mask], dim=0), batch_size)
        if self.fused and self.concurrent_decoders:
            if self._streams is None or self._streams[0] is None:
                self._streams = (ops.extra_stream("dec_a"), ops.side_streams()["dec_b"])
            cen, den = _FusedDecoderPair.apply(tokens, self._packed, self._stack_base["cen"], self._stack_base["den"],
                                               2 * len(self.decoder_centroid_blocks), layouts, self.pos_table,
                                               self.nhead[0], self._streams)
            return cen, den
        cen = self._run_stack(self.decoder_centroid_blocks, "cen", tokens, pos, layouts)
        den = self._run_stack(self.decoder_density_blocks, "den", tokens, pos, layouts)
        return cen, den

    def forward_decoder(self, visible_voxel_feat, coors, coors_mask, batch_size):
        masked_start_id = coors.shape[0]
        cen, den = self.decode(visible_voxel_feat, coors, coors_mask, batch_size)
        cm = cen[masked_start_id:]
        dm = den[masked_start_id:]
        reg_pred_low = self.decoder_pred_low(cm).view(-1, self.per_sub_voxel_num_low, 3)
        reg_pred_med = self.decoder_pred_med(cm).view(-1, self.per_sub_voxel_num_med, 3)
        reg_pred_top = self.decoder_pred_top(cm)
        nor_low = self.decoder_pred_density_low(dm).view(-1, self.per_sub_voxel_num_low, 3) if self.low else None
        nor_med = self.decoder_pred_density_med(dm).view(-1, self.per_sub_voxel_num_med, 3) if self.med else None
        nor_top = self.decoder_pred_density_top(dm) if self.top else None
        if self.cls_sub_voxel:
            cls_low = self.cls_pred_low(cm).view(-1, self.per_sub_voxel_num_low, 2)
            cls_med = self.cls_pred_med(cm).view(-1, self.per_sub_voxel_num_med, 2)
            return reg_pred_low, reg_pred_med, reg_pred_top, nor_low, nor_med, nor_top, cls_low, cls_med
        return reg_pred_low, reg_pred_med, reg_pred_top, nor_low, nor_med, nor_top
