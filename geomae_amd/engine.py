"""Step-level boundary: the whole pre-training iteration as one C call (csrc/engine.hip, geomae_pretrain_*).

`PretrainEngine` binds a built MultiSubVoxelDynamicVoxelNetSSL (parameters re-homed into FlatParams) to the C engine:
it gathers the device pointers once, allocates ONE workspace tensor, and afterwards a training step is
`submit` (first batch only) + `step` -- two ctypes calls, no Python per kernel, no torch allocator traffic.  The host
side of the reference being replaced is mmcv's runner loop around `forward_train` (ssl.py:126-166) + OptimizerHook.

World size > 1: naiveSyncBN1d's four [2C] all-reduces and the early gradient-segment exchanges are issued from a
hook the engine calls at the right points of its schedule (torch.distributed = RCCL over xGMI on the GPU box)."""
import ctypes

import torch
from torch import distributed as dist

from . import _lib, ops
from ._lib import GeomaePretrainConfig, GeomaePretrainModel, check

HOOK_BN_FWD0, HOOK_BN_FWD1, HOOK_BN_BWD1, HOOK_BN_BWD0, HOOK_GRADS_EARLY, HOOK_GRADS_ENCODER, HOOK_FEAT_MOMENTS0, \
    HOOK_FEAT_MOMENTS1 = range(8)
PHASES = ("vfe_fwd", "layouts_wait", "enc_fwd", "dec_fwd", "heads_loss", "dec_bwd", "enc_bwd", "vfe_bwd_stats",
          "vfe_bwd_layer1", "vfe_bwd_layer0", "vfe_bwd_join", "optimizer")
ERR_WORKSPACE = -3


def supported(model):
    """The engine covers exactly the configuration the fused kernels cover (the mae_sst config)."""
    bb, ve = getattr(model, "backbone", None), getattr(model, "voxel_encoder", None)
    return (bb is not None and ve is not None and getattr(bb, "fused", False) and getattr(ve, "use_fused", True)
            and hasattr(model, "_tcfg") and len(ve.vfe_layers) == 2 and next(model.parameters()).is_cuda)


class PretrainEngine:
    def __init__(self, model, flat, opt, max_norm, world=1, max_points=0, max_pillars=0, mask_draws=0, exchange=None):
        self.lib = _lib.load()
        self.model, self.flat, self.opt, self.world = model, flat, opt, int(world)
        self.max_norm = float(max_norm or 0.0)
        # the distributed schedule (hooks + separate optimizer call); forced at world size 1 to exercise the RCCL path
        self.exchange = (self.world > 1) if exchange is None else bool(exchange)
        self.dev = flat.flat.device
        self.handle = None
        self.pending = None                  # the list object of the batch whose stage 1 is enqueued
        self.max_points, self.max_pillars = int(max_points), int(max_pillars)
        self.on_segment = None               # world > 1: callable(i) starting the exchange of gradient segment i
        self.bn_group = None
        st = ops.side_streams(self.dev)
        self.geo, self.aux = st["geo"], st["dec_b"]
        import os
        if os.environ.get("GEOMAE_ENGINE_SERIAL") == "1":
            # diagnostic: the whole schedule on the caller's stream (every cross-stream wait then refers to an event
            # recorded earlier in the same stream).  What a step costs WITHOUT any overlap between hardware queues --
            # the bound for a box whose stream -> queue mapping defeats the three-stream schedule.
            self.geo = self.aux = torch.cuda.current_stream(self.dev)
        self._hook_c = _lib.PRETRAIN_HOOK(self._hook)
        self._profiler = None
        self._phase_timing = False
        self._opt_steps = int(opt.step_count)
        self._hook_error = None
        # steps begun in this RUN: step i's batch draws mask i + 1 whichever engine object (workspace growth, another
        # batch size, checkpoint resume) consumes it -- the C engine counts its own steps, this restores the total
        self.mask_draws = int(mask_draws)

    # ------------------------------------------------------------------ binding
    def _config(self):
        m, bb, ve = self.model, self.model.backbone, self.model.voxel_encoder
        c = GeomaePretrainConfig()
        c.batch_size, c.num_features = self._B, 5
        ctypes.memmove(ctypes.byref(c.targets), ctypes.byref(m._tcfg), ctypes.sizeof(m._tcfg))
        ctypes.memmove(ctypes.byref(c.window), ctypes.byref(bb._wcfg), ctypes.sizeof(bb._wcfg))
        c.num_heads = bb.nhead[0]
        c.encoder_layers, c.decoder_layers = 2 * len(bb.encoder_blocks), 2 * len(bb.decoder_centroid_blocks)
        c.keep_fraction = 1 - m.random_mask_ratio
        c.mask_seed = int(m.mask_seed)
        c.loss_weights[:] = [float(v) for v in (m.loss_ratio_low_nor, m.loss_ratio_low, m.loss_ratio_med, m.loss_ratio_top,
                                                m.cls_loss_ratio_low, m.cls_loss_ratio_med)]
        c.vfe_voxel_size[:] = [float(ve.vx), float(ve.vy), float(ve.vz)]
        c.vfe_center_offset[:] = [float(ve.x_offset), float(ve.y_offset), float(ve.z_offset)]
        n0 = ve.vfe_layers[0].norm
        c.bn_eps, c.bn_momentum = float(n0.eps), float(n0.momentum)
        c.beta1, c.beta2 = float(self.opt.betas[0]), float(self.opt.betas[1])
        c.adam_eps, c.weight_decay, c.max_grad_norm = float(self.opt.eps), float(self.opt.weight_decay), self.max_norm
        c.world_size = self.world
        c.exchange_always = int(self.exchange)
        from .norm import NaiveSyncBatchNorm1d
        c.sync_bn = int(isinstance(n0, NaiveSyncBatchNorm1d))          # a config with plain 'BN1d' keeps local statistics
        c.vfe_bf16 = int(ve.layer1_bf16)                                # (the detector copied the backbone's compute_dtype)
        return c

    def _model_struct(self):
        bb, ve, f, o = self.model.backbone, self.model.voxel_encoder, self.flat, self.opt
        P = bb._packed
        P.refresh()                                         # builds the descriptor table / structs, packs once
        nl = len(P.layers)
        self._keep = dict(layers=P.weight_array(0, nl), grads=P.grad_array(0, nl))
        m = GeomaePretrainModel()
        m.layers = ctypes.cast(self._keep["layers"], ctypes.c_void_p)
        m.layer_grads = ctypes.cast(self._keep["grads"], ctypes.c_void_p)
        m.head_grads = P.head_grads()
        m.head_w_packed, m.head_bias = P.head_w.data_ptr(), P.head_bias.data_ptr()
        m.pos_table = bb.pos_table.data_ptr()
        if bb.mask_token.grad is None:
            bb.mask_token.grad = torch.zeros_like(bb.mask_token)
        m.mask_token, m.mask_token_grad = bb.mask_token.data_ptr(), bb.mask_token.grad.data_ptr()
        m.pack_desc, m.num_pack_desc, m.pack_max_elems = P.desc.data_ptr(), P.n_desc, 384 * 128
        m.packed, m.pack_aux = P.packed.data_ptr(), P.head_bias.data_ptr()
        l0, l1 = ve.vfe_layers
        m.vfe_w0, m.vfe_w1 = l0.linear.weight.data_ptr(), l1.linear.weight.data_ptr()
        m.vfe_dw0, m.vfe_dw1 = l0.linear.weight.grad.data_ptr(), l1.linear.weight.grad.data_ptr()
        for i, l in enumerate((l0, l1)):
            m.bn_gamma[i], m.bn_beta[i] = l.norm.weight.data_ptr(), l.norm.bias.data_ptr()
            m.bn_dgamma[i], m.bn_dbeta[i] = l.norm.weight.grad.data_ptr(), l.norm.bias.grad.data_ptr()
            m.bn_running_mean[i], m.bn_running_var[i] = l.norm.running_mean.data_ptr(), l.norm.running_var.data_ptr()
            m.bn_num_batches[i] = l.norm.num_batches_tracked.data_ptr()
        m.params, m.grads = f.flat.data_ptr(), f.grad.data_ptr()
        m.exp_avg, m.exp_avg_sq, m.num_params = o.exp_avg.data_ptr(), o.exp_avg_sq.data_ptr(), f.flat.numel()
        nds = list(f.nd_ranges)
        prefix = nds.pop(0)[1] if nds and nds[0][0] == 0 else 0
        assert len(nds) <= 1, "the fused AdamW pass takes a no-decay prefix and one more range"
        m.no_decay_prefix = prefix
        m.no_decay2_start, m.no_decay2_count = (nds[0][0], nds[0][1] - nds[0][0]) if nds else (0, 0)
        if self.exchange:                                               # (allocated even without sync_bn: 3 KB)
            z = lambda n, dt: torch.zeros(n, dtype=dt, device=self.dev)
            self.sync = dict(mom0=z(128, torch.float32), mom1=z(256, torch.float32), bs1=z(256, torch.float64),
                             bs0=z(128, torch.float64))
            m.bn_sync_moments0, m.bn_sync_moments1 = self.sync["mom0"].data_ptr(), self.sync["mom1"].data_ptr()
            m.bn_sync_bsums1, m.bn_sync_bsums0 = self.sync["bs1"].data_ptr(), self.sync["bs0"].data_ptr()
            # layer-0 statistics exchanged one step ahead, as rank-averaged feature moments (include/geomae_hip.h
            # FEAT_MOMENTS; GEOMAE_BN0_AHEAD=0: in line, inside the VFE forward)
            import os
            if os.environ.get("GEOMAE_BN0_AHEAD", "1") != "0":
                self.sync["featmom"] = z(2 * 144, torch.float64)
                m.bn_sync_feat_moments = self.sync["featmom"].data_ptr()
        return m

    def _create(self, n_points):
        """(Re)build the C engine for batches of up to max_points / max_pillars (grown when a batch does not fit)."""
        if self.handle is not None:
            torch.cuda.synchronize(self.dev)
            self.lib.geomae_pretrain_destroy(ctypes.c_void_p(self.handle))
            self.handle = None
        self.max_points = max(self.max_points, int(1.25 * n_points) + 1024)
        if self.max_pillars <= 0:
            self.max_pillars = self.max_points // 2
        self._cfg, self._mdl = self._config(), self._model_struct()
        wsb = self.lib.geomae_pretrain_workspace_bytes(ctypes.byref(self._cfg), self.max_points, self.max_pillars)
        if wsb < 0:
            raise _lib.GeomaeLibraryError("geomae_pretrain_workspace_bytes: bad configuration")
        self.ws = torch.empty(wsb, dtype=torch.uint8, device=self.dev)
        streams = (ctypes.c_void_p * 2)(self.geo.cuda_stream, self.aux.cuda_stream)
        h = self.lib.geomae_pretrain_create(ctypes.byref(self._cfg), ctypes.byref(self._mdl), ops._ptr(self.ws), wsb,
                                            self.max_points, self.max_pillars, streams)
        if not h:
            raise _lib.GeomaeLibraryError("geomae_pretrain_create: " + self.lib.geomae_last_error().decode())
        self.handle = h
        self.pending = None
        check(self.lib.geomae_pretrain_set_optimizer_steps(ctypes.c_void_p(h), self._opt_steps), "set_optimizer_steps")
        check(self.lib.geomae_pretrain_set_mask_draws(ctypes.c_void_p(h), self.mask_draws), "set_mask_draws")
        if self.exchange:
            check(self.lib.geomae_pretrain_set_hook(ctypes.c_void_p(h), self._hook_c, None), "geomae_pretrain_set_hook")
        if self._profiler:
            self.set_profiler(self._profiler)
        if self._phase_timing:
            self.set_phase_timing(True)
        self._gnorm = self._view(1, 1)

    def _view(self, what, n, dtype=torch.float32):
        off = self.lib.geomae_pretrain_result_offset(ctypes.c_void_p(self.handle), what)
        assert off >= 0
        return self.ws[off:off + n * torch.empty((), dtype=dtype).element_size()].view(dtype)

    # ------------------------------------------------------------------ hooks (world > 1)
    def _hook(self, user, what, stream):
        # an exception escaping a ctypes callback is only printed: keep it and re-raise after the C call returned
        try:
            self._hook_body(what, stream)
        except BaseException as e:                       # noqa: BLE001
            if self._hook_error is None:
                self._hook_error = e

    def _stream_of(self, handle):
        """The torch stream object for the HIP stream the engine hands to a hook.  RCCL collectives are STREAM-ordered
        (the communicator's stream waits for torch's current stream at the call, and a blocking collective makes the
        current stream wait for its end): the hook must make the stream it was given current, not rely on the caller's
        current stream happening to be that one."""
        handle = int(handle or 0)
        for s in (self.geo, self.aux):
            if s.cuda_stream == handle:
                return s
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream == handle:
            return cur
        return torch.cuda.ExternalStream(handle, device=self.dev)

    # measurement (bench.py, world > 1): event pairs around the collectives the MAIN stream waits for -- the in-line SyncBN
    # all-reduces (the feature-moment exchange and the early gradient segments run on side streams, beside compute)
    comm_events = None                                  # a list while measuring: (what, start event, end event)

    def _hook_body(self, what, stream):
        group = self.bn_group if self.bn_group is not None else ops.BN_GROUP
        inline = what in (HOOK_BN_FWD0, HOOK_BN_FWD1, HOOK_BN_BWD1, HOOK_BN_BWD0) and self.comm_events is not None
        if inline:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self._stream_of(stream))
        self._hook_collective(what, stream, group)
        if inline:
            e1.record(self._stream_of(stream))
            self.comm_events.append((what, e0, e1))

    def _hook_collective(self, what, stream, group):
        with torch.cuda.stream(self._stream_of(stream)):
            if what == HOOK_BN_FWD0:
                dist.all_reduce(self.sync["mom0"], group=group)
            elif what == HOOK_BN_FWD1:
                dist.all_reduce(self.sync["mom1"], group=group)
            elif what == HOOK_BN_BWD1:
                dist.all_reduce(self.sync["bs1"], group=group)
            elif what == HOOK_BN_BWD0:
                dist.all_reduce(self.sync["bs0"], group=group)
            elif what in (HOOK_FEAT_MOMENTS0, HOOK_FEAT_MOMENTS1):
                slot = what - HOOK_FEAT_MOMENTS0
                dist.all_reduce(self.sync["featmom"][144 * slot:144 * (slot + 1)], group=group)
            elif self.on_segment is not None:               # `stream`: the one behind which the segment is complete
                self.on_segment(0 if what == HOOK_GRADS_EARLY else 1)

    # ------------------------------------------------------------------ stepping
    @staticmethod
    def _frames(points):
        B = len(points)
        ptrs, sizes = (ctypes.c_void_p * B)(), (ctypes.c_int64 * B)()
        for i, p in enumerate(points):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 2 and p.shape[1] == 5):
                raise RuntimeError("points must be contiguous CUDA float32 [N_i, 5] tensors")
            ptrs[i], sizes[i] = p.data_ptr(), p.shape[0]
        return ptrs, sizes, sum(int(p.shape[0]) for p in points)

    def _ensure(self, points):
        n = sum(int(p.shape[0]) for p in points)
        if self.handle is None or len(points) != self._B or n > self.max_points:
            self._B = len(points)
            self._create(n)

    def submit(self, points, moments_exchanged=False):
        self._B = getattr(self, "_B", len(points))
        self._ensure(points)
        ptrs, sizes, _ = self._frames(points)
        check(self.lib.geomae_pretrain_submit_ex(ctypes.c_void_p(self.handle), ptrs, sizes, int(bool(moments_exchanged)),
                                                 ops._stream()), "geomae_pretrain_submit")
        self.pending, self._pending_keep = points, (ptrs, sizes)

    def set_mask(self, ids_keep, ids_mask):
        """Replace the random mask of the pending batch by the caller's pillar ids (the reference's
        get_vanilla_mask_index output, ssl.py:287-304): parity tests run the ENGINE on the reference's own mask."""
        ik = ids_keep.to(device=self.dev, dtype=torch.int32).contiguous()
        im = ids_mask.to(device=self.dev, dtype=torch.int32).contiguous()
        check(self.lib.geomae_pretrain_set_mask(ctypes.c_void_p(self.handle), ops._ptr(ik), ik.numel(), ops._ptr(im),
                                                im.numel(), ops._stream()), "geomae_pretrain_set_mask")
        self._mask_keep = (ik, im)                       # read by copies enqueued on the current stream

    def step(self, points, next_points, lr, run_optimizer=True, ids_keep=None, ids_mask=None):
        """One iteration on `points` (submitted now unless it is the batch handed over as the previous step's
        next_points).  ids_keep / ids_mask: an injected mask for `points` instead of the random one.
        -> (losses [6] view, gnorm 0-d view)."""
        if self.handle is None or self.pending is not points:
            self.submit(points)
        if ids_keep is not None:
            self.set_mask(ids_keep, ids_mask)
        nxt = None
        if next_points is not None:
            if sum(int(p.shape[0]) for p in next_points) > self.max_points or len(next_points) != self._B:
                next_points = None                      # does not fit this engine: submitted (and re-sized) next step
            else:
                nxt = self._frames(next_points)
                for t in next_points:                   # read by copies enqueued on the decoder-B stream during this step
                    t.record_stream(self.aux)
        for attempt in range(4):
            rc = self.lib.geomae_pretrain_step(ctypes.c_void_p(self.handle), nxt[0] if nxt else None, nxt[1] if nxt else None,
                                               float(lr), 1.0 / self.world, int(bool(run_optimizer)), ops._stream())
            if self._hook_error is not None:
                err, self._hook_error = self._hook_error, None
                raise RuntimeError("geomae_pretrain_step: a collective issued from the engine's hook failed; the step's "
                                   "results are invalid") from err
            if rc != ERR_WORKSPACE:
                break
            # more pillars than the workspace was sized for: nothing was enqueued (and no hook raised); grow and resubmit.
            # The growth is a decision of THIS rank: the batch's feature moments were exchanged when it was first
            # submitted, a second all-reduce would have no partner -- carry the exchanged moments over instead.
            slot = self.lib.geomae_pretrain_pending_slot(ctypes.c_void_p(self.handle))
            fm = getattr(self, "sync", {}).get("featmom") if self.exchange else None
            if fm is not None:
                # the batch may have been handed over as the PREVIOUS step's next_points: its feature-moment all-reduce
                # was then issued on the decoder-B (aux) stream and nothing on the current stream is ordered behind it yet
                # (the main stream waits for it inside the step that just refused to run).  A clone taken before the
                # reduction landed would carry this rank's LOCAL moments into the re-submission.
                torch.cuda.synchronize(self.dev)
            keep = fm[144 * slot:144 * (slot + 1)].clone() if (fm is not None and slot >= 0) else None
            self.max_pillars = int(1.5 * self.max_pillars) + 1024
            self._create(sum(int(p.shape[0]) for p in points))
            if keep is not None and "featmom" in self.sync:
                self.sync["featmom"][:144].copy_(keep)          # a fresh engine submits into slot 0
                self.submit(points, moments_exchanged=True)
            else:
                self.submit(points)
            if ids_keep is not None:
                self.set_mask(ids_keep, ids_mask)
        check(rc, "geomae_pretrain_step")
        self.mask_draws += 1
        if run_optimizer:
            self._opt_steps += 1
        self.pending, self._pending_keep = next_points, nxt
        return self._view(0, 6), self._gnorm[0]

    def optimizer_step(self, lr):
        check(self.lib.geomae_pretrain_optimizer(ctypes.c_void_p(self.handle), float(lr), 1.0 / self.world, ops._stream()),
              "geomae_pretrain_optimizer")
        self._opt_steps += 1
        return self._gnorm[0]

    # ------------------------------------------------------------------ misc
    def invalidate_packed(self):
        if self.handle is not None:
            check(self.lib.geomae_pretrain_invalidate_packed(ctypes.c_void_p(self.handle)), "invalidate_packed")

    def set_optimizer_steps(self, n):
        self._opt_steps = int(n)
        if self.handle is not None:
            check(self.lib.geomae_pretrain_set_optimizer_steps(ctypes.c_void_p(self.handle), int(n)), "set_optimizer_steps")

    def set_mask_draws(self, n):
        """Steps begun in this run (Trainer.iter after a checkpoint resume): must precede the next batch's submission."""
        self.mask_draws = int(n)
        if self.handle is not None:
            check(self.lib.geomae_pretrain_set_mask_draws(ctypes.c_void_p(self.handle), int(n)), "set_mask_draws")
            self.pending = None                          # a batch drawn under the old counter is re-submitted

    def set_profiler(self, handle):
        self._profiler = handle
        if self.handle is not None:
            check(self.lib.geomae_pretrain_set_profiler(ctypes.c_void_p(self.handle), ctypes.c_void_p(handle) if handle else None),
                  "geomae_pretrain_set_profiler")

    def set_phase_timing(self, on):
        self._phase_timing = bool(on)
        if self.handle is not None:
            check(self.lib.geomae_pretrain_set_phase_timing(ctypes.c_void_p(self.handle), int(bool(on))), "set_phase_timing")

    def phase_times(self):
        """{phase: ms} of the last step (blocks until it has run); needs set_phase_timing(True) before the step."""
        buf = (ctypes.c_float * 16)()
        n = self.lib.geomae_pretrain_phase_times(ctypes.c_void_p(self.handle), buf, 16)
        return {PHASES[i]: float(buf[i]) for i in range(n)}

    def host_times(self):
        """(seconds inside geomae_pretrain_step, seconds of that blocked on the count readback, steps), cumulative."""
        out = (ctypes.c_double * 3)()
        if self.handle is None:                          # no step yet
            return 0.0, 0.0, 0
        check(self.lib.geomae_pretrain_host_times(ctypes.c_void_p(self.handle), out), "geomae_pretrain_host_times")
        return float(out[0]), float(out[1]), int(out[2])

    STACK_FORMS = {-1: "none", 0: "three_launch", 1: "one_launch", 2: "looping"}

    def last_forms(self):
        """Which kernel form each layer stack of the last step took (include/geomae_hip.h GEOMAE_STACK_FORM_*):
        {'enc_fwd': 'one_launch', 'den_fwd': ..., 'cen_fwd': ..., 'enc_bwd': ..., 'den_bwd': ..., 'cen_bwd': ...}."""
        out = (ctypes.c_int32 * 6)()
        n = self.lib.geomae_pretrain_step_forms(ctypes.c_void_p(self.handle), out, 6)
        if n != 6:
            check(n if n < 0 else -1, "geomae_pretrain_step_forms")
        names = ("enc_fwd", "den_fwd", "cen_fwd", "enc_bwd", "den_bwd", "cen_bwd")
        return {k: self.STACK_FORMS.get(int(v), str(int(v))) for k, v in zip(names, out)}

    def last_sizes(self):
        out = (ctypes.c_int64 * 9)()
        n = self.lib.geomae_pretrain_last_sizes_n(ctypes.c_void_p(self.handle), out, 9)
        if n != 9:
            check(n if n < 0 else -1, "geomae_pretrain_last_sizes_n")
        return dict(N=out[0], V=out[1], n_keep=out[2], n_mask=out[3], optimizer_steps=out[4], mask_draws=out[5],
                    max_window_keep=(out[6], out[7]), big_bundle_layouts=out[8])

    def last_ids(self):
        s = self.last_sizes()
        return self._view(2, s["n_keep"], torch.int32), self._view(3, s["n_mask"], torch.int32)

    def last_voxel_coors(self):
        """[V, 4] int32 (b, z, y, x) of the last step's batch, in pillar order (the order ids_keep / ids_mask index)."""
        return self._view(4, 4 * self.last_sizes()["V"], torch.int32).view(-1, 4)

    def close(self):
        if self.handle is not None:
            torch.cuda.synchronize(self.dev)
            self.lib.geomae_pretrain_destroy(ctypes.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
