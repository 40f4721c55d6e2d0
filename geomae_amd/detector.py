"""MultiSubVoxelDynamicVoxelNetSSL -- the GeoMAE pre-training detector (orchestration, random
masking, geometric targets, six-term loss).

Reference: mmdet3d/models/detectors/multi_sub_voxel_dynamic_voxelnet_ssl.py (forward_train :126-166,
extract_feat :169-242, forward_loss :837-902).  Constructor kwargs are exactly those of
configs/mae_sst/*.py (:60-161), including the ones the path never uses (hard_sub_voxel_layer_*,
loss, nor_usr_sml1, ...).  The data flow is re-designed around one counting sort per batch:

  points (list[B] of [N_i,5]) -> cat -> voxelize_batch3 (3 resolutions, 1 kernel)
    -> pillar_segment (replaces 6x torch.unique(dim=0)) -> [the single D2H readback: pillars/sample]
    -> DynamicScatterVFE (segment mean / max kernels) -> random_mask (device)
    -> geometry_targets (centroids, occupancy, normal+curvature: 2 kernels, masked rows only)
    -> MultiMAESSTSPChoose (CSR windows + MFMA window attention) -> forward_loss
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .registry import DETECTORS, build_backbone, build_loss, build_voxel_encoder


@DETECTORS.register_module()
class MultiSubVoxelDynamicVoxelNetSSL(nn.Module):
    def __init__(self, loss, loss_ratio_low, loss_ratio_med, loss_ratio_top, loss_ratio_low_nor,
                 loss_ratio_med_nor, loss_ratio_top_nor, hard_sub_voxel_layer_low, hard_sub_voxel_layer_med,
                 hard_sub_voxel_layer_top, random_mask_ratio, grid_size, sub_voxel_ratio_low, sub_voxel_ratio_med,
                 voxel_layer, sub_voxel_layer_low, sub_voxel_layer_med, voxel_encoder, backbone,
                 spatial_shape=[1, 468, 468], nor_usr_sml1=None, cls_loss_ratio_low=None, cls_loss_ratio_med=None,
                 vis=False, cls_sub_voxel=False, normalize_sub_voxel=None, use_focal_mask=None, norm_curv=True,
                 mse_loss=None, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None):
        super().__init__()
        if neck is not None or bbox_head is not None:
            raise NotImplementedError("the pre-training detector has no neck / bbox head")
        if use_focal_mask is not None:
            raise NotImplementedError("use_focal_mask is dead code in the reference config (None)")
        if not mse_loss or nor_usr_sml1 is not None or not normalize_sub_voxel or not cls_sub_voxel or not norm_curv:
            raise NotImplementedError("only the mae_sst configuration (mse_loss, normalize_sub_voxel, "
                                      "cls_sub_voxel, norm_curv) is implemented")
        self.spatial_shape = spatial_shape
        self.nor_usr_sml1, self.norm_curv = nor_usr_sml1, norm_curv
        self.loss_ratio_low, self.loss_ratio_med, self.loss_ratio_top = loss_ratio_low, loss_ratio_med, loss_ratio_top
        self.loss_ratio_low_nor, self.loss_ratio_med_nor, self.loss_ratio_top_nor = \
            loss_ratio_low_nor, loss_ratio_med_nor, loss_ratio_top_nor
        self.cls_loss_ratio_low, self.cls_loss_ratio_med = cls_loss_ratio_low, cls_loss_ratio_med
        self.cls_sub_voxel, self.vis = cls_sub_voxel, vis
        self.random_mask_ratio = random_mask_ratio
        self.point_cloud_range = voxel_layer["point_cloud_range"]
        self.voxel_size = voxel_layer["voxel_size"]
        self.grid_size = grid_size
        self.sub_voxel_size_low = sub_voxel_layer_low["voxel_size"]
        self.sub_voxel_size_med = sub_voxel_layer_med["voxel_size"]
        self.sub_voxel_ratio_low, self.sub_voxel_ratio_med = sub_voxel_ratio_low, sub_voxel_ratio_med
        self.sub_voxel_layer_low = ops.Voxelization(**sub_voxel_layer_low)
        self.sub_voxel_layer_med = ops.Voxelization(**sub_voxel_layer_med)
        self.hard_sub_voxel_layer_low = ops.Voxelization_with_flag(**hard_sub_voxel_layer_low)
        self.hard_sub_voxel_layer_med = ops.Voxelization_with_flag(**hard_sub_voxel_layer_med)
        self.hard_sub_voxel_layer_top = ops.Voxelization(**hard_sub_voxel_layer_top)
        self.voxel_layer = ops.Voxelization(**voxel_layer)
        self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        self.backbone = build_backbone(backbone)
        # one compute mode for the whole step: the voxel encoder's layer-1 GEMMs follow the backbone's compute_dtype
        # (voxel_encoder.py DynamicScatterVFE.compute_dtype; a stand-alone encoder keeps the fp32-grade products)
        if hasattr(self.voxel_encoder, "compute_dtype") and getattr(self.backbone, "compute_dtype", None) in ("bf16", "fp32"):
            self.voxel_encoder.compute_dtype = self.backbone.compute_dtype
        self.reg_loss = build_loss(loss)
        self.mse_loss, self.use_focal_mask, self.normalize_sub_voxel = mse_loss, use_focal_mask, normalize_sub_voxel
        self.cls_loss = build_loss(dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0))
        self._tcfg = ops.make_target_config(grid_size, sub_voxel_ratio_low, sub_voxel_ratio_med, self.voxel_size,
                                            self.sub_voxel_size_med, self.sub_voxel_size_low, self.point_cloud_range)
        self.mask_seed = 0
        self._iter = 0

    # ------------------------------------------------------------------ mmdet Base3DDetector surface
    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        raise NotImplementedError("the SSL detector is train-only")

    def init_weights(self):
        pass

    def forward_train(self, points, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_bboxes_ignore=None,
                      ids_keep=None, ids_mask=None):
        if getattr(self.backbone, "fused", False):
            return self.forward_train_fused(points, ids_keep=ids_keep, ids_mask=ids_mask)
        x, tgt = self.extract_feat(points, gt_bboxes_3d, gt_labels_3d, img_metas, ids_keep=ids_keep,
                                   ids_mask=ids_mask)
        reg_low, reg_med, reg_top, nor_low, nor_med, nor_top, cls_low, cls_med = x
        return self.forward_loss(tgt["centroid_low"], tgt["mask_low"], tgt["centroid_med"], tgt["mask_med"],
                                 tgt["centroid_top"], tgt["normal"], None, None, reg_low, reg_med, reg_top, nor_low,
                                 nor_med, nor_top, cls_low, cls_med)

    LOSS_KEYS = ("loss_curv_around", "loss_centroid_low", "loss_centroid_med", "loss_centroid_top", "loss_cls_low",
                 "loss_cls_med")

    OVERLAP_GEOMETRY = True     # mask / targets / window CSR on a side stream, concurrent with the VFE forward

    def forward_train_fused(self, points, ids_keep=None, ids_mask=None):
        """Same result as extract_feat + backbone heads + forward_loss, without materialising the eight
        prediction tensors: the heads, the six losses and their gradients are one kernel.

        Everything that depends only on the pillar coordinates -- the random mask, the geometric targets and
        the four window layouts (a dozen launches of one to a few hundred workgroups, ~0.3 ms of a mostly idle
        GPU) -- is enqueued on a side stream while the main stream runs the VFE forward."""
        batch_size = len(points)
        voxels, coors, sub_med, sub_low, seg = self._stage1(points)
        V = seg.V                                                     # the iteration's one host readback
        main = torch.cuda.current_stream()
        overlap = self.OVERLAP_GEOMETRY
        if overlap:
            side = ops.side_streams()["geo"]
            side.wait_stream(main)
        else:
            side = main
        with torch.cuda.stream(side), torch.no_grad():
            if ids_keep is None:
                ik, im, token_row, counts = self.get_vanilla_mask_index(seg)
            else:
                ik, im = ids_keep.int(), ids_mask.int()
                token_row, counts = ops.token_rows_from_ids(ik, im, V)
            tgt = ops.geometry_targets(voxels, seg, sub_med, sub_low, self._tcfg, token_row, counts,
                                       n_rows=int(im.numel()))
            ik, im = ik.long(), im.long()
            feature_coors = seg.voxel_coors[:V]
            coors_keep, coors_mask = feature_coors[ik], feature_coors[im]
            layouts = self.backbone.build_layouts(coors_keep, coors_mask, batch_size)
        voxel_features, _ = self.voxel_encoder(voxels, coors, seg=seg)
        if overlap:
            main.wait_stream(side)
        w = (self.loss_ratio_low_nor, self.loss_ratio_low, self.loss_ratio_med, self.loss_ratio_top,
             self.cls_loss_ratio_low, self.cls_loss_ratio_med)
        losses = self.backbone.forward_losses(voxel_features[ik], coors_keep, coors_mask, batch_size, tgt, w,
                                              layouts=layouts)
        self._loss_vector = losses                   # Trainer sums this once instead of adding six scalars
        return {k: losses[i] for i, k in enumerate(self.LOSS_KEYS)}

    @torch.no_grad()
    def train_step_explicit(self, points, on_early_grads=None, next_points=None, on_encoder_grads=None):
        """forward_train_fused + backward as an explicit schedule: no autograd tape or engine.  Accumulates every
        parameter gradient into .grad and returns the loss dict (detached).  Used by Trainer for the fused path
        (the step was host-bound at ~3.3 ms of Python / autograd per 3.5 ms of GPU work)."""
        batch_size = len(points)
        voxels, coors, sub_med, sub_low, seg = self._stage1(points)
        V = seg.V
        main = torch.cuda.current_stream()
        streams = ops.side_streams()
        side, aux = streams["geo"], streams["dec_b"]
        step_end, self._step_end_event = getattr(self, "_step_end_event", None), None
        if step_end is not None:        # recorded by the trainer after its optimizer step; the decoder-B stream already
            side.wait_event(step_end)   # waited for it (pack-ahead), and nothing was enqueued on the main stream since
        else:
            side.wait_stream(main)
            aux.wait_stream(main)
        dev = voxels.device
        C = self.backbone.mask_token.shape[1]
        f32 = torch.float32
        # Everything that depends only on the points and the pillar coordinates runs beside the VFE forward, on two
        # streams, arranged so that the main stream waits for other queues as rarely as possible (a cross-queue wait
        # costs ~15 us of queue time even when its event fired long ago, tools/archive/phase_events.py):
        #   geometry stream : (random mask, unless `prefetch` drew it in the previous step) -> token coordinates -> the
        #                     four window layouts.  ONE event (layouts_ready) gates the encoder; the weights packed
        #                     ahead by the trainer are chained into it.
        #   decoder-B stream: idle until the decoders fork, so it first takes the step's zero arena, the NEXT batch's
        #                     stage 1 (voxelize / pillar sort / count readback / VFE front / random mask) and the
        #                     geometric targets.  The main stream joins it after the decoder forward anyway: no wait of
        #                     its own for any of them (the targets are first read by the heads+loss kernel).
        with torch.cuda.stream(side):
            ops.mark("side:start")
            packed_ready = self.backbone._packed.refresh_if_stale()   # a no-op when the trainer packed after its optimizer step
            ik, im, token_row, counts = self.get_vanilla_mask_index(seg)
            mask_done = side.record_event()
            ops.mark("side:mask")
            n_keep, n_mask = int(ik.numel()), int(im.numel())
            n = n_keep + n_mask
            coors_all, _ = ops.gather_token_coors(ik, im, seg.voxel_coors)
            ops.mark("side:coors")
            layouts = self.backbone.build_layouts(coors_all[:n_keep], None, batch_size, coors_all=coors_all)
            ops.mark("side:layouts")
            if packed_ready is not None:
                side.wait_event(packed_ready)
            layouts_ready = side.record_event()
        ops.mark("step_start")
        # the VFE forward's own arena is one fill on the main stream: a cross-queue wait costs as much as the fill
        prepared, self._prepared_points = self._prepared_points, None
        zeros_fwd = ops.ZeroArena(ops.ZeroArena.nbytes(*ops.vfe_forward_zero_specs(seg.cap, V, prepared is not None)), dev)
        vf, vfe_state = self.voxel_encoder.forward_explicit(voxels, seg, zeros=zeros_fwd, prepared=prepared)
        ops.mark("vfe_fwd_done")
        vfe_done = main.record_event()
        with torch.cuda.stream(aux):
            # behind the VFE forward: its sweeps fill the chip, the encoder that follows leaves 150 of 256 CUs idle
            aux.wait_event(vfe_done)
            late = [((n, C), f32), ((n, C), f32), ((n, C), f32), ((V, C), f32), ((6,), f32)] + ops.vfe_backward_zero_specs(V)
            zeros_late = ops.ZeroArena(ops.ZeroArena.nbytes(*late), dev)
            bufs = dict(d_cen=zeros_late.take((n, C), f32), d_cen2=zeros_late.take((n, C), f32), d_den=zeros_late.take((n, C), f32),
                        d_vf=zeros_late.take((V, C), f32), losses=zeros_late.take((6,), f32), side=side)
        if next_points is not None:
            self.prefetch(next_points, stream=aux)
        with torch.cuda.stream(aux):
            aux.wait_event(mask_done)
            tgt = ops.geometry_targets(voxels, seg, sub_med, sub_low, self._tcfg, token_row, counts, n_rows=n_mask)
            ops.mark("side:targets")
        main.wait_event(layouts_ready)
        ops.mark("layouts_awaited")
        w = (self.loss_ratio_low_nor, self.loss_ratio_low, self.loss_ratio_med, self.loss_ratio_top,
             self.cls_loss_ratio_low, self.cls_loss_ratio_med)
        # the gather of the kept pillars and the scatter of their gradients (ids_keep are distinct rows: masked pillars
        # get no gradient) are folded into the encoder's first / last kernel
        losses, d_vf = self.backbone.losses_and_grads_explicit(vf, n_mask, batch_size, tgt, w, layouts, on_early_grads,
                                                               packed_fresh=True, bufs=bufs, keep_rows=ik,
                                                               on_encoder_grads=on_encoder_grads)
        join = bufs.get("join_side", False)          # work parked on the geometry stream: only the optimizer needs it
        self.voxel_encoder.backward_explicit(vfe_state, d_vf, zeros=zeros_late, side=side if join else None)
        if join:
            main.wait_stream(side)
        ops.mark("vfe_bwd_done")
        return {k: losses[i] for i, k in enumerate(self.LOSS_KEYS)}

    # ------------------------------------------------------------------ preprocessing
    @torch.no_grad()
    def voxelize_all(self, points):
        """voxelize + sub_voxelize_low + sub_voxelize_med (ssl.py:307-377) in one kernel."""
        sizes = [int(p.shape[0]) for p in points]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        pts = torch.cat([p.float() for p in points], dim=0).contiguous() if len(points) > 1 else points[0].float().contiguous()
        # pinned + non_blocking: a pageable host->device copy blocks the CPU until every kernel enqueued before
        # it has run (it was a hidden full sync per step)
        boffs = torch.tensor(offs, dtype=torch.int32).pin_memory().to(pts.device, non_blocking=True)
        top, med, low = ops.voxelize_batch3(pts, boffs, len(points), self.voxel_size, self.sub_voxel_size_med,
                                            self.sub_voxel_size_low, self.point_cloud_range)
        return pts, top, med, low

    def voxelize(self, points):
        pts, top, _, _ = self.voxelize_all(points)
        return pts, top

    @torch.no_grad()
    def _launch_mask(self, seg):
        self._iter += 1
        # (window-major token lists when the backbone runs the fused stacks: csrc/mask.hip)
        return ops.random_mask_launch(seg, 1 - self.random_mask_ratio, (self.mask_seed << 32) + self._iter,
                                      getattr(self.backbone, "_wcfg", None))

    @torch.no_grad()
    def get_vanilla_mask_index(self, seg):
        raw, self._mask_raw = getattr(self, "_mask_raw", None), None
        if raw is None or raw[0] is not seg:           # not drawn ahead by `prefetch` for this batch
            raw = (seg, self._launch_mask(seg))
        return ops.random_mask_finish(raw[1], seg)

    @torch.no_grad()
    def prefetch(self, points, stream=None):
        """Stage 1 of `prepare` for a FUTURE batch: voxelize x3 + pillar segments are enqueued now and the
        pillar counts travel to pinned host memory asynchronously.  Call it for batch k+1 before enqueuing
        step k (Trainer.train_step(next_points=...)): when step k+1 starts the counts are already on the host,
        so the iteration's one device->host readback no longer drains the queue (the GPU idled ~0.3 ms per
        step behind it, profiles/r01q_step_timeline.txt).
        stream: enqueue there without further ordering (the explicit schedule puts it on the decoder-B stream, which
        idles until the decoders fork; a fifth stream would share a hardware queue with another one)."""
        main = torch.cuda.current_stream()
        if stream is not None:
            ps = stream
        else:
            ps = getattr(self, "_prefetch_stream", None) or ops.prefetch_stream()
            # its own stream: nothing in the current step depends on it.  Ordered after the work already enqueued on
            # the main stream (the previous step), which also makes the allocator's reuse of this stream's freed
            # blocks safe: their last readers (the previous step's kernels) are enqueued on `main` before this point.
            ps.wait_stream(main)
        with torch.cuda.stream(ps):
            voxels, coors, sub_med, sub_low = self.voxelize_all(points)
            seg = ops.pillar_segment(coors, len(points), self.grid_size)
            seg.start_readback()
            # the weight-independent front of the fused VFE (pillar means, decorated features in pillar order) rides
            # along: ~45 us of dependent small kernels that would otherwise open the step's critical path
            prepared = self.voxel_encoder.prepare_points(voxels, seg) if getattr(self.voxel_encoder, "use_fused", True) \
                and hasattr(self.voxel_encoder, "prepare_points") else None
            # ... and so does the random mask of that batch: it needs the per-sample pillar offsets on the DEVICE only
            # (one draw per batch, in batch order: the same sequence of seeds as drawing inside each step)
            mask_raw = self._launch_mask(seg)
            done = ps.record_event()
        # stream given = the explicit schedule's decoder-B stream: the main stream joins that stream later in the same
        # step (after the decoder forward), so the consumer of this batch needs no wait of its own
        self._prefetched = (points, (voxels, coors, sub_med, sub_low, seg), done, prepared, stream is not None, mask_raw)

    def _stage1(self, points):
        """voxelize x3 + pillar segments: taken from `prefetch` when it ran for this batch."""
        pre, self._prefetched = getattr(self, "_prefetched", None), None
        self._prepared_points = None
        if pre is not None and pre[0] is points:
            if not (len(pre) > 4 and pre[4]):                # (on the geometry stream: ordered by its later events, see prefetch)
                torch.cuda.current_stream().wait_event(pre[2])
            self._prepared_points = pre[3] if len(pre) > 3 else None
            self._mask_raw = (pre[1][4], pre[5]) if len(pre) > 5 else None
            return pre[1]
        voxels, coors, sub_med, sub_low = self.voxelize_all(points)
        seg = ops.pillar_segment(coors, len(points), self.grid_size)
        return voxels, coors, sub_med, sub_low, seg

    def prepare(self, points, ids_keep=None, ids_mask=None):
        """voxelize x3 -> pillar segments -> VFE -> mask -> geometric targets (ssl.py:172-231)."""
        voxels, coors, sub_med, sub_low, seg = self._stage1(points)
        V = seg.V                                                     # the iteration's one host readback
        voxel_features, feature_coors = self.voxel_encoder(voxels, coors, seg=seg)
        if ids_keep is None:
            ids_keep, ids_mask, token_row, counts = self.get_vanilla_mask_index(seg)
        else:
            ids_keep, ids_mask = ids_keep.int(), ids_mask.int()
            token_row, counts = ops.token_rows_from_ids(ids_keep, ids_mask, V)
        with torch.no_grad():
            tgt = ops.geometry_targets(voxels, seg, sub_med, sub_low, self._tcfg, token_row, counts,
                                       n_rows=int(ids_mask.numel()))
        return voxel_features, feature_coors, ids_keep.long(), ids_mask.long(), tgt

    def extract_feat(self, points, gt_bboxes_3d=None, gt_labels_3d=None, img_metas=None, vis=False, ids_keep=None,
                     ids_mask=None):
        voxel_features, feature_coors, ik, im, tgt = self.prepare(points, ids_keep, ids_mask)
        x = self.backbone(voxel_features[ik], feature_coors[ik], feature_coors[im], len(points))
        return x, tgt

    # ------------------------------------------------------------------ losses (ssl.py:837-902)
    def forward_loss(self, centroid_low, centroid_low_mask, centroid_med, centroid_med_mask, centroid_high,
                     centroid_normal_low, centroid_normal_med, centroid_normal_high, reg_pred_low, reg_pred_med,
                     reg_pred_high, nor_pred_low, nor_pred_med, nor_pred_high, cls_pred_low=None, cls_pred_med=None):
        def masked_mse(pred, tgt, mask, ratio):
            # == ((pred - tgt)[mask] ** 2).mean(-1).sum() / mask.sum(): no boolean indexing -> no host sync
            m = mask.reshape(-1)
            per = ((pred.reshape(-1, 3).float() - tgt.reshape(-1, 3)) ** 2).mean(dim=-1)
            return (per * m).sum() / m.sum().clamp(min=1).to(per.dtype) * ratio

        def mse(pred, tgt, ratio):
            per = ((pred.float() - tgt) ** 2).mean(dim=-1)
            return per.sum() / per.shape[0] * ratio

        loss_reg_low = masked_mse(reg_pred_low, centroid_low, centroid_low_mask, self.loss_ratio_low)
        loss_reg_med = masked_mse(reg_pred_med, centroid_med, centroid_med_mask, self.loss_ratio_med)
        loss_reg_top = mse(reg_pred_high, centroid_high, self.loss_ratio_top)
        nor_pred = nor_pred_high if (nor_pred_low is None and nor_pred_med is None) else nor_pred_low
        loss_nor_low = mse(nor_pred, centroid_normal_low, self.loss_ratio_low_nor)
        loss_cls_low = self.cls_loss(cls_pred_low.reshape(-1, 2).float(), centroid_low_mask.reshape(-1).long()) \
            * self.cls_loss_ratio_low
        loss_cls_med = self.cls_loss(cls_pred_med.reshape(-1, 2).float(), centroid_med_mask.reshape(-1).long()) \
            * self.cls_loss_ratio_med
        return dict(loss_curv_around=loss_nor_low, loss_centroid_low=loss_reg_low, loss_centroid_med=loss_reg_med,
                    loss_centroid_top=loss_reg_top, loss_cls_low=loss_cls_low, loss_cls_med=loss_cls_med)
