"""naiveSyncBN1d -- the reference's cross-rank batch norm (mmdet3d/ops/norm.py:9-86), the only
collective call site in the tree besides DDP.

Semantics kept exactly: per-rank mean and mean-of-squares are averaged with EQUAL weight per rank
(norm.py:70), var = E[x^2] - E[x]^2, running stats updated with the biased variance (norm.py:73-76);
with world_size == 1 (or in eval) it is plain nn.BatchNorm1d (norm.py:58-59).  Forward uses
all_gather and backward all_reduce of a [2C] fp32 vector (RCCL over xGMI on the GPU box, gloo in the
CPU tests) -- host logic, device agnostic.
"""
import torch
from torch import distributed as dist
from torch import nn
from torch.autograd.function import Function

from .registry import NORM_LAYERS


class AllReduce(Function):
    @staticmethod
    def forward(ctx, input):
        input_list = [torch.zeros_like(input) for _ in range(dist.get_world_size())]
        dist.all_gather(input_list, input, async_op=False)
        return torch.sum(torch.stack(input_list, dim=0), dim=0)

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        dist.all_reduce(grad_output, async_op=False)
        return grad_output


@NORM_LAYERS.register_module("naiveSyncBN1d")
class NaiveSyncBatchNorm1d(nn.BatchNorm1d):
    def forward(self, input):
        assert input.dtype == torch.float32, f"input should be in float32 type, got {input.dtype}"
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or not self.training:
            return super().forward(input)
        assert input.shape[0] > 0, "SyncBN does not support empty inputs"
        assert input.dim() == 2
        C = input.shape[1]
        mean = torch.mean(input, dim=0)
        meansqr = torch.mean(input * input, dim=0)
        vec = torch.cat([mean, meansqr], dim=0)
        vec = AllReduce.apply(vec) * (1.0 / dist.get_world_size())
        mean, meansqr = torch.split(vec, C)
        var = meansqr - mean * mean
        with torch.no_grad():
            self.running_mean += self.momentum * (mean.detach() - self.running_mean)
            self.running_var += self.momentum * (var.detach() - self.running_var)
        invstd = torch.rsqrt(var + self.eps)
        scale = self.weight * invstd
        bias = self.bias - mean * scale
        return input * scale.reshape(1, -1) + bias.reshape(1, -1)


@NORM_LAYERS.register_module("naiveSyncBN2d")
class NaiveSyncBatchNorm2d(nn.BatchNorm2d):
    """mmdet3d/ops/norm.py:89-140: the 4-D variant used by the fine-tune backbone's conv stack and SECONDFPN."""

    def forward(self, input):
        assert input.dtype == torch.float32, f"input should be in float32 type, got {input.dtype}"
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or not self.training:
            return super().forward(input)
        assert input.shape[0] > 0, "SyncBN does not support empty inputs"
        C = input.shape[1]
        mean = torch.mean(input, dim=[0, 2, 3])
        meansqr = torch.mean(input * input, dim=[0, 2, 3])
        vec = AllReduce.apply(torch.cat([mean, meansqr], dim=0)) * (1.0 / dist.get_world_size())
        mean, meansqr = torch.split(vec, C)
        var = meansqr - mean * mean
        with torch.no_grad():
            self.running_mean += self.momentum * (mean.detach() - self.running_mean)
            self.running_var += self.momentum * (var.detach() - self.running_var)
        invstd = torch.rsqrt(var + self.eps)
        scale = self.weight * invstd
        bias = self.bias - mean * scale
        return input * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


NORM_LAYERS.register_module("BN1d", module=nn.BatchNorm1d)
NORM_LAYERS.register_module(["BN", "BN2d"], module=nn.BatchNorm2d)
