"""The two loss types the mae_sst config names (built by the detector, ssl.py:106,112-115).

CrossEntropyLoss(use_sigmoid=True) restates mmdet 2.20.0's sigmoid branch (un-vendored in the
reference): [K, C] logits vs [K] labels -> one-hot -> BCE-with-logits averaged over K*C elements,
times loss_weight.  SmoothL1Loss is built by the config but unused (mse_loss=True)."""
import torch.nn.functional as F
from torch import nn

from .registry import LOSSES


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, loss_weight=1.0):
        super().__init__()
        if not use_sigmoid:
            raise NotImplementedError("only use_sigmoid=True is on the pre-training path")
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight

    def forward(self, cls_score, label, **kwargs):
        onehot = F.one_hot(label, cls_score.shape[-1]).to(cls_score.dtype)
        return self.loss_weight * F.binary_cross_entropy_with_logits(cls_score, onehot, reduction=self.reduction)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, **kwargs):
        return self.loss_weight * F.smooth_l1_loss(pred, target, beta=self.beta, reduction=self.reduction)
