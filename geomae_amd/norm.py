"""naiveSyncBN1d / naiveSyncBN2d -- the reference's cross-rank batch norm as ONE closed-form autograd node.

Behaviour being matched (mmdet3d/ops/norm.py:28-86, 89-140; the only collective call site of the tree besides
DDP): in training with world_size > 1 the per-rank mean and mean of squares of every channel are averaged with EQUAL
weight per rank (whatever the ranks' point counts), var = E[x^2] - E[x]^2, the running statistics move towards
(mean, biased var) with `momentum` and num_batches_tracked is left alone; otherwise the module is its torch base
class.  tests/golden/g_syncbn_w2.npz holds what the reference module itself produces at world size 2.

Mechanism (not the reference's): statistics and gradient are computed by hand in `_CrossRankNorm` -- one [2C]
all-reduce forward, one backward, fp64 partial sums like the fused VFE kernels (csrc/vfe.hip bn_finalize /
vfe_bwd_*), no autograd graph through the reductions.  With y = x * s + t, s = gamma * r, t = beta - m * s,
r = (q - m^2 + eps)^-1/2 and (m, q) the rank-averaged moments:
    d beta = sum dy            d gamma = r * (sum dy x - m sum dy)                    (local sums)
    g_q = -0.5 * gamma * r^3 * (sum dy x - m sum dy)        g_m = -s * sum dy - 2 m g_q
    dx = dy * s + (G_m + 2 x G_q) / (world * n_local),   (G_m, G_q) = all-reduce of (g_m, g_q).
Device agnostic (gloo on CPU in the tests, RCCL on the GPU box); the pre-training hot path does not run this module
at all -- its VFE computes the same statistics inside the fused sweeps (ops.vfe_forward / csrc/engine.hip)."""
import torch
from torch import distributed as dist
from torch import nn

from .registry import NORM_LAYERS


def _cross_rank_active(module):
    return module.training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _CrossRankNorm(torch.autograd.Function):
    """x [N, C, *] fp32 -> y; channel axis 1; returns (y, mean, var) with the moments detached."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, group):
        world = dist.get_world_size(group)
        C = x.shape[1]
        red = [d for d in range(x.dim()) if d != 1]
        shape = [1, C] + [1] * (x.dim() - 2)
        n_local = x.numel() // C
        xd = x.double()
        mom = torch.cat([xd.sum(red), (xd * xd).sum(red)]).div_(n_local).float()      # local (mean, mean of squares)
        dist.all_reduce(mom, group=group)
        mom.div_(world)                                                              # equal weight per rank
        m, q = mom[:C], mom[C:]
        var = q - m * m
        r = torch.rsqrt(var + eps)
        s = gamma * r
        y = x * s.view(shape) + (beta - m * s).view(shape)
        ctx.save_for_backward(x, gamma, m, r)
        ctx.group, ctx.world, ctx.n_local, ctx.red, ctx.shape = group, world, n_local, red, shape
        ctx.mark_non_differentiable(m, var)
        return y, m, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, gamma, m, r = ctx.saved_tensors
        C = x.shape[1]
        dyd = dy.double()
        sum_dy = dyd.sum(ctx.red)
        sum_dyx = (dyd * x.double()).sum(ctx.red)
        centred = sum_dyx - m.double() * sum_dy
        d_beta, d_gamma = sum_dy.float(), (r.double() * centred).float()
        g_q = -0.5 * gamma.double() * r.double() ** 3 * centred
        g_m = -(gamma.double() * r.double()) * sum_dy - 2.0 * m.double() * g_q
        G = torch.cat([g_m, g_q]).float()
        dist.all_reduce(G, group=ctx.group)
        k = 1.0 / (ctx.world * ctx.n_local)
        dx = dy * (gamma * r).view(ctx.shape) + (G[:C].view(ctx.shape) + 2.0 * x * G[C:].view(ctx.shape)) * k
        return dx, d_gamma, d_beta, None, None


class _CrossRankMixin:
    """forward() shared by the 1-d and 2-d variants; `process_group` (None = default) may be set by the owner."""
    process_group = None

    def forward(self, input):
        if input.dtype != torch.float32:
            raise AssertionError(f"naiveSyncBN expects float32 activations, got {input.dtype}")
        if not _cross_rank_active(self):
            return super().forward(input)
        if input.shape[0] == 0:
            raise AssertionError("naiveSyncBN cannot normalise an empty batch (every rank must contribute statistics)")
        y, mean, var = _CrossRankNorm.apply(input, self.weight, self.bias, self.eps, self.process_group)
        with torch.no_grad():
            self.running_mean.lerp_(mean, self.momentum)
            self.running_var.lerp_(var, self.momentum)
        return y


@NORM_LAYERS.register_module("naiveSyncBN1d")
class NaiveSyncBatchNorm1d(_CrossRankMixin, nn.BatchNorm1d):
    """[N, C] or [N, C, L] activations (mmdet3d/ops/norm.py:28-86)."""


@NORM_LAYERS.register_module("naiveSyncBN2d")
class NaiveSyncBatchNorm2d(_CrossRankMixin, nn.BatchNorm2d):
    """[N, C, H, W] activations: the fine-tune backbone's conv stack and SECONDFPN (mmdet3d/ops/norm.py:89-140)."""


NORM_LAYERS.register_module("BN1d", module=nn.BatchNorm1d)
NORM_LAYERS.register_module(["BN", "BN2d"], module=nn.BatchNorm2d)
