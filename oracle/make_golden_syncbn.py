"""Generate tests/golden/g_syncbn_w2.npz by running the REFERENCE's naiveSyncBN1d at world size 2 (build container only).

TEST INFRASTRUCTURE ONLY.  Two gloo ranks import mmdet3d/ops/norm.py (NaiveSyncBatchNorm1d, ops/norm.py:28-86) and the
reference's DynamicScatterVFE (voxel_encoder.py:308-419) from /root/reference under the stub modules of
oracle/ref_import.py; the VFE's norm layers are the REAL cross-rank module.  Only inputs / outputs are written.
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_syncbn.py

Part 1 (module): x_r [N_r, 8] with N_0 = 37, N_1 = 53 -> y_r, dx_r, local d gamma / d beta, running stats.
Part 2 (A3 + A4): the whole VFE on two different pairs of frames -> voxel_feats (every 4th row + column sums), the
local parameter gradients of loss_r = sum(vf_r * w_r), and the running statistics (equal on both ranks)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
TOP = (0.256, 0.256, 8)


def frames_of(rank):
    """The frames tests/test_gpu_multirank.py feeds rank `rank`."""
    from geomae_amd import synth
    return [synth.lidar_frame(300 + 10 * rank + i, beams=16, n_az=300 + 40 * rank) for i in range(2)]


def module_inputs(rank):
    g = torch.Generator().manual_seed(40 + rank)
    n = (37, 53)[rank]
    return torch.randn(n, 8, generator=g) * 2.0 + 0.5, torch.randn(n, 8, generator=g)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(4)
    import ref_import
    import geomae_oracle as O
    ref = ref_import.load_reference()
    norm_mod = ref_import._load("mmdet3d.ops.norm", "mmdet3d/ops/norm.py")
    SyncBN = norm_mod.NaiveSyncBatchNorm1d
    out = {}
    # ---- part 1: the module alone
    bn = SyncBN(8, eps=1e-3, momentum=0.01).train()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 8))
        bn.bias.copy_(torch.linspace(-0.2, 0.3, 8))
    x, w = module_inputs(rank)
    xin = x.clone().requires_grad_(True)
    y = bn(xin * 1.0)                                    # (the module unsqueezes its input in place: give it a non-leaf)
    (y * w).sum().backward()
    out.update(m_y=y.detach().numpy(), m_dx=xin.grad.numpy(), m_dgamma=bn.weight.grad.numpy(), m_dbeta=bn.bias.grad.numpy(),
               m_running_mean=bn.running_mean.numpy(), m_running_var=bn.running_var.numpy(),
               m_num_batches_tracked=np.int64(bn.num_batches_tracked.item()))
    # ---- part 2: DynamicScatterVFE with the real cross-rank norm
    utils_mod = sys.modules["mmdet3d.models.voxel_encoders.utils"]
    utils_mod.build_norm_layer = lambda cfg, c: ("bn", SyncBN(c, eps=cfg["eps"], momentum=cfg["momentum"]))
    vfe = ref.vfe.DynamicScatterVFE(in_channels=5, feat_channels=[64, 128], with_distance=False, voxel_size=TOP,
                                    with_cluster_center=True, with_voxel_center=True, point_cloud_range=RANGE,
                                    norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01))
    assert type(vfe.vfe_layers[0].norm) is SyncBN
    params = O.make_params(7, 1, 1)
    vsd = {k[len("voxel_encoder."):]: v for k, v in params.items() if k.startswith("voxel_encoder.")}
    vfe.load_state_dict(vsd, strict=False)
    vfe.train()
    frames = frames_of(rank)
    cs = []
    for i, p in enumerate(frames):
        pts = torch.as_tensor(p)
        c = pts.new_zeros((pts.shape[0], 3), dtype=torch.int32)
        ref.voxel_layer.dynamic_voxelize(pts, c, list(map(float, TOP)), list(map(float, RANGE)), 3)
        cs.append(F.pad(c, (1, 0), value=i))
    voxels, coors = torch.cat([torch.as_tensor(p) for p in frames], 0), torch.cat(cs, 0)
    vf, vc = vfe(voxels, coors)
    wv = torch.randn(vf.shape, generator=torch.Generator().manual_seed(rank))
    (vf * wv).sum().backward()
    out.update(v_n_points=np.int64(voxels.shape[0]), v_coors=vc.numpy().astype(np.int16), v_rows=vf.detach()[::4].numpy(),
               v_colsum=vf.detach().double().sum(0).numpy(), v_abssum=vf.detach().double().abs().sum(0).numpy())
    for k, p in vfe.named_parameters():
        out["v_grad." + k] = p.grad.numpy()
    for k, b in vfe.named_buffers():
        out["v_buf." + k] = b.numpy()
    np.savez(os.path.join(tmp, f"r{rank}.npz"), **out)
    dist.destroy_process_group()


def main():
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        merged = {}
        for r in range(2):
            d = np.load(os.path.join(tmp, f"r{r}.npz"))
            for k in d.files:
                merged[f"r{r}.{k}"] = d[k]
    dst = os.path.join(os.environ.get("GEOMAE_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "g_syncbn_w2.npz")
    np.savez_compressed(dst, **merged)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: v.shape for k, v in merged.items() if k.startswith("r0.") and v.ndim})
    assert np.array_equal(merged["r0.v_buf.vfe_layers.1.norm.running_var"], merged["r1.v_buf.vfe_layers.1.norm.running_var"])


if __name__ == "__main__":
    main()
