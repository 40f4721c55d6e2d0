"""Anchor3DHead (fine-tune config) in plain torch vs the reference's own head run under stubs
(tests/golden/g_head.npz, oracle/make_golden_head.py): anchors, target assignment, the three losses and every gradient.
The head is torch-only by design (SURVEY: "leave to MIOpen/PyTorch"), so this runs on CPU.

What the fixture pins: the reference's own Anchor3DHead / train_mixins / anchor generator / box coder code.  What it
cannot pin: mmdet 2.20's MaxIoUAssigner, FocalLoss, SmoothL1Loss and bbox_overlaps, which are absent from the tree and
the image -- the generator ran the reference head with the SAME restatements of them standing in (make_golden_head.py), so
for target assignment and the loss formulas this is a self-consistency check ("parity unpinned")."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))


def _setup():
    from make_golden_head import HEAD_CFG, TRAIN_CFG, inputs          # config dicts + seeded inputs (no reference import)
    from geomae_amd.dense_head import Anchor3DHead
    head = Anchor3DHead(train_cfg=dict(TRAIN_CFG), test_cfg=None, **HEAD_CFG)
    sd = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(sum(map(ord, k)))) * (0.05 if v.dim() > 1 else 0.5)
          for k, v in head.state_dict().items()}
    head.load_state_dict(sd)
    return head, inputs()


def test_anchor3d_head_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g_head.npz"))
    head, (feat, gts, labels) = _setup()
    anchors = head.anchor_generator.grid_anchors([feat.shape[-2:]], device="cpu")[0]
    assert anchors.shape[0] == int(g["num_anchors"]) and head.num_anchors == 14
    assert np.array_equal(anchors[::997].numpy(), g["anchors_sample"])
    t = head.targets_single(anchors, gts[0], labels[0])
    assert np.array_equal(t[0].numpy().astype(np.int8), g["labels0"])
    assert np.array_equal(t[4].numpy().astype(np.int8), g["dir_targets0"])
    assert np.allclose(t[2][t[3].sum(-1) > 0].numpy(), g["bbox_targets_pos0"], rtol=1e-6, atol=1e-6)
    x = feat.clone().requires_grad_(True)
    losses = head.forward_train([x], None, gts, labels)
    assert set(losses) == {"loss_cls", "loss_bbox", "loss_dir"} and all(len(v) == 1 for v in losses.values())
    for k in losses:
        assert abs(float(losses[k][0].detach()) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), (k, float(losses[k][0]), float(g[k]))
    sum(sum(v) for v in losses.values()).backward()
    assert np.allclose(x.grad.double().sum(dim=(0, 2, 3)).numpy(), g["dx_sum"], rtol=1e-4, atol=1e-5)
    assert abs(float(x.grad.double().abs().sum()) - float(g["dx_abs"])) <= 1e-4 * float(g["dx_abs"])
    for k, p in head.named_parameters():
        ref = g["grad." + k]
        assert np.linalg.norm(p.grad.numpy() - ref) <= 1e-4 * max(np.linalg.norm(ref), 1e-6), k


def test_anchor3d_head_without_ground_truth_and_config_errors():
    head, (feat, gts, labels) = _setup()
    empty = [torch.zeros((0, 9)) for _ in gts]
    losses = head.forward_train([feat], None, empty, [torch.zeros(0, dtype=torch.long) for _ in gts])
    assert float(losses["loss_bbox"][0]) == 0.0 and float(losses["loss_dir"][0]) == 0.0 and torch.isfinite(losses["loss_cls"][0])
    from geomae_amd.dense_head import Anchor3DHead
    with pytest.raises(NotImplementedError):
        Anchor3DHead(10, 8, anchor_generator=dict(type="Anchor3DRangeGenerator", ranges=[[0, 0, 0, 1, 1, 1]]))
    with pytest.raises(NotImplementedError):
        Anchor3DHead(10, 8, assigner_per_size=True)


def test_max_iou_assigner_semantics():
    """pos >= 0.6, neg < 0.3, in between ignored (-1); every gt keeps its best anchor if that reaches min_pos_iou."""
    from geomae_amd.dense_head import max_iou_assign
    ov = torch.tensor([[0.7, 0.2, 0.45, 0.10, 0.35],
                       [0.1, 0.25, 0.5, 0.05, 0.35]])
    a = max_iou_assign(ov, 0.6, 0.3, 0.3)
    assert a.tolist() == [1, 0, 2, 0, -1]         # anchor 2: best anchor of gt 1 (0.5 >= min_pos_iou); anchor 4: 0.35 is neither
    a = max_iou_assign(ov, 0.6, 0.3, 0.55)
    assert a.tolist() == [1, 0, -1, 0, -1]


def test_vectorised_low_quality_match_equals_the_sequential_loop():
    """max_iou_assign's match_low_quality step without a Python loop over the gts (one host sync per gt before): per anchor
    the LARGEST claiming gt index, which is what mmdet's in-order loop leaves behind ("a later gt overrides")."""
    from geomae_amd.dense_head import max_iou_assign

    def loop(overlaps, pos, neg, minp, assign_all):
        G, A = overlaps.shape
        assigned = overlaps.new_full((A,), -1, dtype=torch.long)
        max_ov, argmax_ov = overlaps.max(0)
        assigned[(max_ov >= 0) & (max_ov < neg)] = 0
        p = max_ov >= pos
        assigned[p] = argmax_ov[p] + 1
        gt_max = overlaps.max(1).values
        for i in range(G):
            if gt_max[i] >= minp:
                if assign_all:
                    assigned[overlaps[i] == gt_max[i]] = i + 1
                else:
                    assigned[overlaps[i].argmax()] = i + 1
        return assigned
    g = torch.Generator().manual_seed(0)
    for _ in range(100):
        G, A = int(torch.randint(1, 9, (1,), generator=g)), int(torch.randint(5, 60, (1,), generator=g))
        ov = (torch.rand(G, A, generator=g) * 10).round() / 10            # coarse values: many ties
        for assign_all in (True, False):
            assert torch.equal(max_iou_assign(ov, 0.6, 0.45, 0.45, True, assign_all), loop(ov, 0.6, 0.45, 0.45, assign_all))
