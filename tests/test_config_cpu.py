"""Registry / config host logic (CPU)."""
import os

import pytest
import torch

import geomae_amd
from geomae_amd import Config
from geomae_amd.configs import mae_sst_model

REF_CFG = "/root/reference/configs/mae_sst/m_sst_nus_singlestage_curv_07_ssl_dataset_wo_dbsampler_6x_1e-5.py"


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="reference not mounted (GPU box)")
def test_reference_config_loads_unchanged_and_matches_restated_dict():
    cfg = Config.fromfile(REF_CFG)
    assert cfg.model.type == "MultiSubVoxelDynamicVoxelNetSSL"
    assert cfg.optimizer.type == "AdamW" and cfg.optimizer_config.grad_clip.max_norm == 10
    assert cfg.data.samples_per_gpu == 4 and cfg.runner.max_epochs == 72
    assert _plain(cfg.model) == _plain(mae_sst_model())
    model = geomae_amd.build_model(cfg.model)
    assert sum(p.numel() for p in model.parameters()) == 2743382 + 17472


def test_state_dict_keys_match_reference_names():
    import geomae_oracle as O
    model = geomae_amd.build_model(mae_sst_model())
    keys = {k for k, _ in model.named_parameters()}
    assert keys == {n for n, _ in O.param_shapes(6, 2)}
    for n, shape in O.param_shapes(6, 2):
        assert tuple(model.state_dict()[n].shape) == shape
    assert "backbone.encoder_blocks.5.encoder_list.1.win_attn.self_attn.in_proj_weight" in keys


def test_registry_errors():
    with pytest.raises(KeyError):
        geomae_amd.build_model(dict(type="NoSuchDetector"))
    with pytest.raises(KeyError):
        geomae_amd.build_norm_layer(dict(type="NoSuchNorm"), 8)
    name, layer = geomae_amd.build_norm_layer(dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01), 8)
    assert layer.eps == 1e-3 and layer.momentum == 0.01


def test_config_inheritance_and_delete(tmp_path):
    (tmp_path / "base.py").write_text("a = dict(x=1, y=dict(z=2, w=3))\nb = 5\n")
    (tmp_path / "child.py").write_text("_base_ = ['base.py']\na = dict(y=dict(_delete_=True, q=9))\nc = [1, 2]\n")
    cfg = Config.fromfile(str(tmp_path / "child.py"))
    assert cfg.a.x == 1 and dict(cfg.a.y) == {"q": 9} and cfg.b == 5 and cfg.c == [1, 2]
    cfg.merge_from_dict({"a.x": 7, "d.e": 1})
    assert cfg.a.x == 7 and cfg.d.e == 1


def test_pos_table_matches_oracle():
    import geomae_oracle as O
    from geomae_amd.sst import pos_embed_table
    assert torch.equal(pos_embed_table((12, 12), 128), O.pos_embed_table((12, 12), 128))


FT_CFG = "/root/reference/configs/pre_sst/m_sst_nus_second_pointpillar_fpn355_222_curv_07_ssl_data_wo_dbsampler_6x_1e-5.py"


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


@pytest.mark.skipif(not os.path.exists(FT_CFG), reason="reference not mounted (GPU box)")
def test_finetune_config_builds_and_takes_pretrained_encoder():
    """N1: the fine-tune config file resolves unchanged -- detector, middle encoder, backbone, neck AND the Anchor3DHead
    of its base config -- it equals the restated dict the GPU tests use, and the pre-trained backbone.encoder_blocks.*
    keys load into it."""
    from geomae_amd.configs import pre_sst_model
    cfg = Config.fromfile(FT_CFG)
    m = dict(cfg.model)
    assert m["type"] == "DynamicVoxelNet" and m["middle_encoder"]["type"] == "SSTInputLayer"
    assert m["backbone"]["type"] == "SSTSecondPretrainedv1" and m["neck"]["type"] == "SECONDFPN"
    assert _plain(m) == _plain(pre_sst_model())
    model = geomae_amd.build_model(m)
    assert len(model.backbone.encoder_blocks) == 6 and len(model.backbone.conv_blocks) == 3
    head = model.bbox_head
    assert type(head).__name__ == "Anchor3DHead" and head.num_anchors == 14 and head.box_code_size == 9
    assert head.conv_cls.out_channels == 140 and head.conv_reg.out_channels == 126 and head.conv_dir_cls.out_channels == 28
    pre = geomae_amd.build_model(mae_sst_model())
    src = {k: v for k, v in pre.state_dict().items() if k.startswith("backbone.encoder_blocks.")}
    missing = model.load_state_dict(src, strict=False)
    assert not missing.unexpected_keys
    assert not [k for k in missing.missing_keys if k.startswith("backbone.encoder_blocks.")]
    a = pre.state_dict()["backbone.encoder_blocks.3.encoder_list.1.linear1.weight"]
    assert torch.equal(model.state_dict()["backbone.encoder_blocks.3.encoder_list.1.linear1.weight"], a)


@pytest.mark.skipif(not os.path.exists(FT_CFG), reason="reference not mounted (GPU box)")
def test_every_pretrain_and_finetune_config_of_the_reference_builds():
    """configs/mae_sst/*.py and configs/pre_sst/*.py (6 files) load unchanged through Config.fromfile + build_model;
    the two CenterHead fine-tune configs build everything but their head (CenterHead is mmdet3d machinery outside 8(f))."""
    import glob
    root = os.path.dirname(os.path.dirname(FT_CFG))
    files = sorted(glob.glob(os.path.join(root, "mae_sst", "*.py")) + glob.glob(os.path.join(root, "pre_sst", "*.py")))
    assert len(files) == 6
    for f in files:
        model = geomae_amd.build_model(dict(Config.fromfile(f).model))
        name = type(model).__name__
        assert name == ("MultiSubVoxelDynamicVoxelNetSSL" if "mae_sst" in f else "DynamicVoxelNet"), f
        if "pointpillar" in f:
            assert type(model.bbox_head).__name__ == "Anchor3DHead"
