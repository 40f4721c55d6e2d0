"""Training-step plumbing for the pre-training hot path: flat parameter / gradient buffers, the
data-parallel gradient exchange (one RCCL all-reduce over xGMI per step), gradient clipping and
AdamW with the config's `norm` no-decay rule.

Reference behaviour being matched (external to its tree, mmcv/mmdet): MMDistributedDataParallel
(bucketed all-reduce, gradients AVERAGED over ranks), OptimizerHook(grad_clip=dict(max_norm=10,
norm_type=2)), AdamW(lr=1e-5, betas=(0.9,0.999), weight_decay=0.05, paramwise_cfg custom_keys
{'norm': decay_mult 0}) -- configs/_base_/schedules/cosine_2x.py:1-17.

Design for one node of 8 MI355X: the model has 2.76 M parameters (11 MB fp32), i.e. ONE bucket.
Parameters and gradients live in two flat fp32 buffers (every nn.Parameter is a view), so the
exchange is a single in-place all_reduce of the gradient buffer with no flatten/unflatten copies,
clipping is one norm over one buffer, and the optimizer is one fused update over two segments
(decayed / undecayed).  Host logic is device agnostic (tested with gloo on CPU, world_size 2).
"""
import ctypes

import os
import torch
from torch import distributed as dist


class FlatParams:
    """Re-homes every parameter of `model` (and its .grad) into two contiguous fp32 buffers.

    Order: up to two SEGMENTS, each laid out [undecayed ('norm' in the name, or listed in no_decay_keys) | decayed] and
    padded to a multiple of 4 elements.  Segment 0 holds the parameters whose gradients are complete EARLY in the
    backward pass (decoders, heads), segment 1 those matched by `late_keys` (encoder, voxel encoder): the trainer
    all-reduces segment 0 while the encoder backward is still running.  Without late_keys there is one segment."""

    def __init__(self, model, no_decay_keys=("norm",), late_keys=()):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        is_nd = lambda n: any(k in n for k in no_decay_keys)
        # late_keys: name prefixes, or tuples of prefixes, one entry per later segment (in the order their gradients
        # become complete during the backward); everything else forms segment 0
        late_groups = [(k,) if isinstance(k, str) else tuple(k) for k in late_keys]
        if late_groups and all(isinstance(k, str) for k in late_keys):
            late_groups = [tuple(late_keys)]                        # a flat tuple of prefixes = ONE late segment
        which = lambda n: next((i + 1 for i, ks in enumerate(late_groups) if any(n.startswith(k) for k in ks)), 0)
        groups = [[(n, p) for n, p in named if which(n) == gi] for gi in range(len(late_groups) + 1)]
        groups = [g for g in groups if g]
        self.names, self.params, self.offsets, self.segments, self.nd_ranges = [], [], [], [], []
        model_names = [n for n, _ in named]
        off = 0
        for gi, g in enumerate(groups):
            nd = [(n, p) for n, p in g if is_nd(n)]
            dc = [(n, p) for n, p in g if not is_nd(n)]
            # segments alternate [no-decay | decay], [decay | no-decay], ...: the no-decay parts of segments 1 and 2 are
            # then adjacent and the whole buffer has at most two no-decay ranges for up to three segments (the fused
            # AdamW pass takes a prefix range and one more)
            nd_first = gi % 2 == 0
            start = off
            n_nd = sum(p.numel() for _, p in nd)
            n_dc = sum(p.numel() for _, p in dc)
            for n, p in (nd + dc if nd_first else dc + nd):
                self.names.append(n)
                self.params.append(p)
                self.offsets.append(off)
                off += p.numel()
            a = start if nd_first else start + n_dc
            if n_nd:
                if self.nd_ranges and self.nd_ranges[-1][1] == a:
                    self.nd_ranges[-1] = (self.nd_ranges[-1][0], a + n_nd)
                else:
                    self.nd_ranges.append((a, a + n_nd))
            end = (off + 3) // 4 * 4                                # keep every segment 16-byte aligned
            if not nd_first and n_nd and end != off:
                self.nd_ranges[-1] = (self.nd_ranges[-1][0], end)   # the padding (zeros, zero gradients) joins the range
            off = end
            self.segments.append((start, off, n_nd))
        total = off
        where = {n: j for j, n in enumerate(self.names)}
        self.model_order = [where[n] for n in model_names]      # flat index of the i-th TRAINABLE parameter of the model
        # ... and of the i-th parameter counting the frozen ones too (None): the index space of an mmcv / torch optimizer
        # checkpoint, whose DefaultOptimizerConstructor lists every parameter of the model
        self.all_order = [where.get(n) for n, _ in model.named_parameters()]
        self.all_params = [p for _, p in model.named_parameters()]
        self.n_no_decay = self.segments[0][2] if len(self.segments) == 1 else None
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view_as(p.data)
            p.grad = self.grad[o:o + n].view_as(p.data)
        self.total = total

    def zero_grad(self):
        self.grad.zero_()

    def decay_ranges(self):
        """[(start, end)] of the elements that take weight decay."""
        out, pos = [], 0
        for a, b in self.nd_ranges:
            if a > pos:
                out.append((pos, a))
            pos = b
        if pos < self.total:
            out.append((pos, self.total))
        return out

    def check_storage(self):
        """load_state_dict copies in place, so every parameter must still be a view of the flat buffer."""
        base, item = self.flat.data_ptr(), self.flat.element_size()
        for n, p, o in zip(self.names, self.params, self.offsets):
            assert p.data_ptr() == base + o * item, f"parameter {n} left the flat buffer"

    def check_views(self):
        """Autograd must have accumulated in place; re-point any .grad that was replaced."""
        base, item = self.grad.data_ptr(), self.grad.element_size()
        for p, o in zip(self.params, self.offsets):
            g, n = p.grad, p.numel()
            if g is None:
                p.grad = self.grad[o:o + n].view_as(p.data)
            elif g.data_ptr() != base + o * item:
                view = self.grad[o:o + n]
                view.copy_(g.reshape(-1))
                p.grad = view.view_as(p.data)


def exchange_mode():
    """(world size, whether the step runs the distributed schedule).  The schedule of world > 1 -- SyncBN exchanges and
    early gradient-segment all-reduces issued from the engine's hooks, the optimizer as a second call -- can be forced
    at world size 1 with GEOMAE_FORCE_EXCHANGE=1 (and an initialised process group): the collectives then run on RCCL
    with one rank, which is how the stream-ordered code path is exercised on a one-GPU box (tests/test_gpu_nccl.py)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return 1, False
    world = dist.get_world_size()
    return world, world > 1 or os.environ.get("GEOMAE_FORCE_EXCHANGE") == "1"


def _grad_group():
    from . import ops
    return ops.GRAD_GROUP


def _wait_all(works):
    """Make the current stream wait for the gradient-segment all-reduces.  On RCCL every collective of one process group
    runs on that group's communication stream, in issue order: waiting for the LAST one orders the current stream behind
    all of them with one cross-queue wait instead of one per segment (each costs ~10 us of queue time on the waiting
    stream, docs/LAB_NOTES.md rounds 1-3).  Other backends (gloo: host threads) are waited for one by one."""
    if not works:
        return
    # The single wait leans on ProcessGroupNCCL's one-communication-stream-per-group behaviour (an implementation detail of
    # today's torch, not a documented contract): GEOMAE_WAIT_LAST_ONLY=0 waits for every work instead (~10 us each).
    import os
    if dist.get_backend(_grad_group()) == "nccl" and os.environ.get("GEOMAE_WAIT_LAST_ONLY", "1") != "0":
        works[-1][1].wait()
    else:
        for _, w in works:
            w.wait()


def _new_comm_group(what):
    """A process group over all ranks for `what`; on RCCL with a high-priority communication stream.  None = use the
    default group (correct, only less overlapped)."""
    try:
        import os
        if dist.get_backend() == "nccl" and os.environ.get("GEOMAE_COMM_PRIORITY", "normal") == "high":
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                return dist.new_group(pg_options=opts)
            except (AttributeError, TypeError):
                pass
        return dist.new_group()
    except Exception as e:
        import warnings
        warnings.warn(f"geomae_amd: no separate process group for the {what} exchange ({e!r}); using the default group")
        return None


def _move_comm_streams_off_main(dev, attempts=4):
    """The step's hooks issue collectives from its SIDE streams (gradient segments: geometry stream; the next batch's
    feature moments: decoder-B stream).  A stream-ordered collective makes the communicator's internal stream wait for the
    issuing stream, and a wait is a barrier packet for everything behind it in the hardware queue: a communicator whose
    stream shares the MAIN stream's queue makes the main stream wait for the side streams -- the two decoder stacks then
    run one after the other (seen once in ~20 forced one-rank runs: 2.61 instead of 2.10 ms).  More streams than queues
    exist, so sharing as such cannot be avoided, only moved: measured here (ops.comm_blocks), and a group found on the
    main stream's queue is re-created after taking one more stream from torch's pool (the communicator's stream is the
    pool's next one, and pool stream k sits on queue k mod 4).  Every rank takes part in new_group: the decision is
    all-reduced.  Recorded in ops.STREAM_PROBE."""
    from . import ops
    key = (dev.type, dev.index)
    info = ops.STREAM_PROBE.setdefault(key, {})
    if dist.get_backend() != "nccl" or os.environ.get("GEOMAE_STREAM_PROBE", "1") == "0":
        return
    st = ops.side_streams(dev)
    main = torch.cuda.current_stream(dev)
    cycles = ops._spin_us(dev)
    alone = min(ops._spin_seconds(main, cycles, dev) for _ in range(2))
    moved = {}
    for name, issuer, helper in (("GRAD_GROUP", st["geo"], st["dec_b"]), ("BN_GROUP", st["dec_b"], st["geo"])):
        for attempt in range(attempts + 1):
            group = getattr(ops, name)
            if group is None:
                break
            dt = min(ops.comm_blocks(main, issuer, helper, group, cycles, dev) for _ in range(2))
            flag = torch.tensor([1.0 if dt > 1.5 * alone else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag.item()) == 0.0 or attempt == attempts:
                moved[name] = dict(recreated=attempt, blocks_main=bool(float(flag.item())))
                break
            keep = torch.cuda.Stream(device=dev)              # one pool slot further
            with torch.cuda.stream(keep):
                torch.zeros(1, device=dev)
            new = dist.new_group()
            t = torch.zeros(1, device=dev)
            dist.all_reduce(t, group=new)
            torch.cuda.synchronize(dev)
            setattr(ops, name, new)                           # (the old communicator stays alive, idle)
    info["comm_streams"] = moved


def allreduce_gradients(flat, group=None):
    """DDP semantics: gradients averaged over ranks; one collective on the single bucket."""
    if group is None:
        from . import ops
        group = ops.GRAD_GROUP
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=group)
    flat.grad.mul_(1.0 / dist.get_world_size(group))


def clip_grad_norm(flat, max_norm, norm_type=2):
    """torch.nn.utils.clip_grad_norm_ on the flat buffer (mmcv OptimizerHook.clip_grads)."""
    assert norm_type == 2
    total = torch.linalg.vector_norm(flat.grad)
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
    flat.grad.mul_(coef)
    return total


class FlatAdamW:
    """AdamW over the flat buffers; weight decay only on the decayed segment.  Update rule of
    torch.optim.AdamW: p *= 1 - lr*wd ; m,v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).

    On a GPU the whole step (gradient norm, clip, update, zero_grad) is two HIP kernels
    (geomae_grad_sumsq + geomae_adamw_step, csrc/optim.hip); `step()` alone keeps the composed torch form
    (CPU / gloo tests, and the parity check of the kernel)."""

    def __init__(self, flat, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.base_lr = lr
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        self._sumsq = self._gnorm = None
        self._sumsq_zeroed = False

    @torch.no_grad()
    def step(self):
        f = self.flat
        self._sumsq_zeroed = False
        self.step_count += 1
        b1, b2 = self.betas
        g = f.grad
        if self.weight_decay != 0:
            for a, b in f.decay_ranges():
                f.flat[a:b].mul_(1 - self.lr * self.weight_decay)
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        denom = (self.exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        f.flat.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)

    @torch.no_grad()
    def fused_clip_step(self, max_norm=0.0, grad_scale=1.0, zero_grad=True):
        """clip_grad_norm_(max_norm) + AdamW (+ zero_grad) in two launches; returns the pre-clip gradient norm
        (0-d tensor).  grad_scale folds the data-parallel 1/world averaging into the same pass."""
        import ctypes
        from . import _lib
        from . import ops
        from .ops import _ptr, _stream
        lib, f = _lib.load(), self.flat
        if self._sumsq is None:
            self._sumsq_ring = torch.zeros(2, dtype=torch.float64, device=f.flat.device)
            self._sumsq_zeroed, self._sumsq_slot = True, 0
            self._gnorm = torch.zeros(1, dtype=torch.float32, device=f.flat.device)
        if not self._sumsq_zeroed:                   # state was loaded / another path stepped: do not trust the ring
            self._sumsq_ring.zero_()
            self._sumsq_zeroed = True
        self.step_count += 1
        n = f.flat.numel()
        # two accumulator slots: this step sums into one (already zero), the other is zeroed by this step's update
        # pass for the next step -- no fill kernel between the last backward kernel and the norm reduction.  The slot
        # has its own toggle (not step_count's parity: load_state_dict / a composed step() change that)
        slot = self._sumsq_slot
        self._sumsq_slot = 1 - slot
        self._sumsq = self._sumsq_ring[slot:slot + 1]
        main = torch.cuda.current_stream()
        with ops.prezeroed():
            _lib.check(lib.geomae_grad_sumsq(_ptr(f.grad), n, _ptr(self._sumsq), _stream()), "geomae_grad_sumsq")
        # one launch over the whole flat buffer: the segments (each [no-decay | decay]) only matter to the gradient
        # exchange; the kernel takes the second segment's no-decay range and clears the other accumulator slot
        segs, nds = f.segments, list(f.nd_ranges)
        assert segs[0][0] == 0 and segs[-1][1] == n
        prefix = nds.pop(0)[1] if nds and nds[0][0] == 0 else 0
        assert len(nds) <= 1, "the fused AdamW pass takes a no-decay prefix and one more range"
        nd2 = (nds[0][0], nds[0][1] - nds[0][0]) if nds else (0, 0)
        nxt = self._sumsq_ring[1 - slot:2 - slot]
        _lib.check(lib.geomae_adamw_step(_ptr(f.flat), _ptr(f.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), n,
                                         prefix, float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                         float(self.eps), float(self.weight_decay), self.step_count,
                                         float(max_norm or 0.0), _ptr(self._sumsq), float(grad_scale), int(bool(zero_grad)),
                                         _ptr(self._gnorm), nd2[0], nd2[1], _ptr(nxt), _stream()), "geomae_adamw_step")
        ops.mark("optimizer_done")
        return self._gnorm[0]

    # ---- torch.optim.AdamW-shaped state (what mmcv's checkpoint hook stores under 'optimizer'): mmcv's
    # DefaultOptimizerConstructor with a paramwise_cfg makes ONE param group per parameter
    def _model_order(self):
        """(param, flat offset) in model.named_parameters() order -- the index mmcv's DefaultOptimizerConstructor and
        torch.optim use for 'state' / 'param_groups' -- whatever the flat buffer's segment layout is."""
        f = self.flat
        return [(f.params[j], f.offsets[j]) for j in f.model_order]

    def _checkpoint_order(self):
        """[(parameter, flat offset or None)] over ALL parameters of the model in named_parameters() order: frozen ones
        (requires_grad=False, e.g. in the fine-tune configs) keep their slot in the index space, with no state."""
        f = self.flat
        return [(p, None if j is None else f.offsets[j]) for p, j in zip(f.all_params, f.all_order)]

    def state_dict(self):
        f, state, groups = self.flat, {}, []
        decayed = f.decay_ranges()
        for i, (p, off) in enumerate(self._checkpoint_order()):
            if off is None:                      # frozen: a parameter group without state, as torch.optim writes it
                groups.append(dict(params=[i], lr=self.lr, initial_lr=self.base_lr, betas=tuple(self.betas), eps=self.eps,
                                   weight_decay=self.weight_decay, amsgrad=False))
                continue
            n = p.numel()
            wd = self.weight_decay if any(a <= off < b for a, b in decayed) else 0.0
            state[i] = dict(step=torch.tensor(float(self.step_count)),
                            exp_avg=self.exp_avg[off:off + n].view_as(p).clone(),
                            exp_avg_sq=self.exp_avg_sq[off:off + n].view_as(p).clone())
            groups.append(dict(params=[i], lr=self.lr, initial_lr=self.base_lr, betas=tuple(self.betas), eps=self.eps,
                               weight_decay=wd, amsgrad=False))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        order = self._checkpoint_order()
        extra = [k for k in sd["state"] if not (isinstance(k, int) and 0 <= k < len(order))]
        if extra:
            raise ValueError(f"optimizer state has entries for parameters this model does not have: {extra[:4]}")
        for i, (p, off) in enumerate(order):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None and off is None:
                continue                         # state of a parameter that is frozen here: nothing to restore
            if st is not None:
                for key in ("exp_avg", "exp_avg_sq"):
                    if tuple(st[key].shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state {i}: {key} has shape {tuple(st[key].shape)}, the parameter "
                                         f"{tuple(p.shape)} (the checkpoint is from a different model or parameter order)")
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
        self._sumsq_zeroed = False                   # the accumulator ring is re-zeroed by the next fused step
        if sd.get("param_groups"):
            self.lr = sd["param_groups"][0]["lr"]
            self.base_lr = sd["param_groups"][0].get("initial_lr", self.base_lr)


class CyclicLr:
    """mmcv CyclicLrUpdaterHook (un-vendored dependency, mmcv 1.x `runner/hooks/lr_updater.py`; parity unpinned:
    restated from its published behaviour) as selected by configs/_base_/schedules/cosine_2x.py:10-15:
    policy='cyclic', target_ratio=(100, 1e-3), cyclic_times=1, step_ratio_up=0.1, cosine annealing, per iteration.
    Phase 1 [0, up): base_lr -> base_lr * ratio_up ; phase 2 [up, max): base_lr * ratio_up -> base_lr * ratio_down."""

    def __init__(self, base_lr, max_iters, target_ratio=(100, 1e-3), cyclic_times=1, step_ratio_up=0.1):
        import math
        self._cos, self._pi = math.cos, math.pi
        self.base_lr = base_lr
        per = max_iters // cyclic_times
        up = int(step_ratio_up * per)
        self.per = max(per, 1)
        self.phases = [(0, up, 1.0, target_ratio[0]), (up, per, target_ratio[0], target_ratio[1])]

    def lr_at(self, it):
        it %= self.per
        for start, end, r0, r1 in self.phases:
            if start <= it < end:
                factor = (it - start) / (end - start)
                a, b = self.base_lr * r0, self.base_lr * r1
                return b + 0.5 * (a - b) * (self._cos(self._pi * factor) + 1)
        return self.base_lr * self.phases[-1][3]


class Trainer:
    """One process per GPU.  train_step = forward_train + backward + gradient exchange + clip + AdamW."""

    def __init__(self, model, optimizer_cfg=None, grad_clip=None, lr_schedule=None):
        from .configs import GRAD_CLIP, OPTIMIZER
        ocfg = dict(optimizer_cfg or OPTIMIZER)
        assert ocfg.pop("type") == "AdamW"
        pw = ocfg.pop("paramwise_cfg", None) or {}
        keys = tuple(k for k, v in pw.get("custom_keys", {}).items() if v.get("decay_mult", 1.0) == 0.0)
        self.model = model
        # gradients of the decoders / heads are complete before the encoder backward starts: they form the early
        # segment, whose all-reduce overlaps the rest of the backward (explicit schedule, world > 1)
        # ... and the encoder's before the voxel encoder's backward: three segments [decoders + heads | encoder | VFE];
        # only the last, small one (70 KB) is exchanged with nothing left to hide it
        late = (("backbone.encoder_blocks.",), ("voxel_encoder.",)) if hasattr(model, "train_step_explicit") else ()
        self.flat = FlatParams(model, no_decay_keys=keys or ("\0",), late_keys=late)
        if late and exchange_mode()[1]:
            from . import ops
            # Communicators of the step's own: one for the gradient segments, one for the SyncBN statistics (they must not
            # queue behind a gradient all-reduce in flight).  (Every rank builds its trainer: new_group is a collective
            # call.)  GEOMAE_COMM_PRIORITY=high gives them high-priority communication streams -- measured on one MI355X
            # with the forced one-rank exchange (tools/fx_envs.sh): 3.64 instead of 2.20 ms per step at the default four
            # hardware queues (the step's side streams lose their queues to the extra priority level), so it is off.
            if ops.BN_GROUP is None:
                ops.BN_GROUP = _new_comm_group("SyncBN")
            if ops.GRAD_GROUP is None:
                ops.GRAD_GROUP = _new_comm_group("gradient")
            if self.flat.flat.is_cuda:
                dev = self.flat.flat.device
                try:        # first use creates the communicators and their streams ...
                    t = torch.zeros(1, device=dev)
                    dist.all_reduce(t, group=ops.GRAD_GROUP, async_op=True).wait()
                    if ops.BN_GROUP is not None:
                        dist.all_reduce(t, group=ops.BN_GROUP)
                    torch.cuda.synchronize(dev)
                except Exception as e:                       # still correct; the probe below then sees fewer streams
                    import warnings
                    warnings.warn(f"geomae_amd: could not prime the communication streams ({e!r})")
                ops.reset_side_streams(dev)                  # ... and only then are the step's side streams chosen
                ops.side_streams(dev)
                _move_comm_streams_off_main(dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # what MMDistributedDataParallel does at construction: every replica starts from rank 0's parameters AND
            # buffers (BatchNorm running statistics, counters) whatever the ranks seeded or loaded; afterwards only
            # gradients (and the SyncBN statistics) are exchanged
            dist.broadcast(self.flat.flat, src=0)
            for b in model.buffers():
                if b.numel():
                    dist.broadcast(b, src=0)
        self.opt = FlatAdamW(self.flat, **ocfg)
        self.grad_clip = dict(GRAD_CLIP if grad_clip is None else grad_clip)
        self.lr_schedule = lr_schedule          # e.g. CyclicLr(base_lr, max_iters): lr set before every step
        self.iter = 0
        self.fused_optimizer = self.flat.flat.is_cuda
        self.explicit_schedule = self.flat.flat.is_cuda
        # the whole step as one C call (geomae_amd/engine.py, csrc/engine.hip); False = the Python explicit schedule
        # (same kernels, same order: kept as the A/B reference of the engine)
        self.use_engine = self.flat.flat.is_cuda
        self.engine = None

    def get_engine(self):
        """The step engine this trainer drives (created on first use), or None when the Python schedule is in charge."""
        from . import engine as _engine
        if not (self.use_engine and self.explicit_schedule and hasattr(self.model, "train_step_explicit")
                and _engine.supported(self.model)):
            return None
        if self.engine is None:
            world, exchange = exchange_mode()
            self.engine = _engine.PretrainEngine(self.model, self.flat, self.opt, self.grad_clip.get("max_norm", 0.0), world,
                                                 mask_draws=self.iter, exchange=exchange)
        return self.engine

    def _engine_step(self, points, next_points, world, exchange):
        from .detector import MultiSubVoxelDynamicVoxelNetSSL as Det
        eng = self.get_engine()
        if not getattr(self, "_grads_clean", False):
            self.flat.zero_grad()
        if self.lr_schedule is not None:
            self.opt.lr = self.lr_schedule.lr_at(self.iter)
        # the engine re-packs the bf16 weight copies after its own optimizer pass; anything else that wrote a parameter
        # since (a torch op on it or on the flat buffer: tensor version counters) invalidates them
        packed = self.model.backbone._packed
        versions = (self.flat.flat._version, packed._versions())
        if getattr(self, "_engine_versions", None) not in (None, versions):
            eng.invalidate_packed()
        works = []
        if exchange:
            def segment_ready(i):
                a, b, _ = self.flat.segments[i]
                works.append((i, dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=_grad_group(), async_op=True)))
            eng.on_segment = segment_ready if len(self.flat.segments) > 2 else None
        losses, gnorm = eng.step(points, next_points, self.opt.lr, run_optimizer=not exchange)
        if exchange:
            tail = getattr(self, "tail_comm_events", None)      # measurement (bench.py): the exposed end of the exchange
            if tail is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            done = {i for i, _ in works}
            for i, (a, b, _) in enumerate(self.flat.segments):
                if i not in done:
                    works.append((i, dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=_grad_group(), async_op=True)))
            _wait_all(works)
            if tail is not None:
                e1.record()
                tail.append((e0, e1))
            tap = getattr(self, "on_reduced_grad", None)
            if tap is not None:
                tap(self.flat.grad)
            gnorm = eng.optimizer_step(self.opt.lr)
        self._grads_clean = True
        self._engine_versions = versions
        self.opt.step_count = eng._opt_steps
        self.iter += 1
        return {k: losses[i] for i, k in enumerate(Det.LOSS_KEYS)}, gnorm

    def train_step(self, points, next_points=None, **kw):
        """next_points: the batch of the FOLLOWING step (the same list object must be passed as `points`
        then); its voxelization / pillar sort is enqueued ahead of this step so that its count readback is
        off the critical path (detector.prefetch)."""
        world, exchange = exchange_mode()
        if self.use_engine and self.explicit_schedule and not kw and hasattr(self.model, "train_step_explicit"):
            from . import engine as _engine
            if _engine.supported(self.model):
                return self._engine_step(points, next_points, world, exchange)
        if not getattr(self, "_grads_clean", False):
            self.flat.zero_grad()
        self._grads_clean = False
        if self.lr_schedule is not None:
            self.opt.lr = self.lr_schedule.lr_at(self.iter)
        if getattr(self, "_prepack_flat_version", None) not in (None, self.flat.flat._version):
            self.model.backbone._packed.invalidate()        # someone wrote the flat parameter buffer since the pre-pack
        self._prepack_flat_version = None
        pre = getattr(self.model, "_prefetched", None)
        if pre is not None and pre[0] is points:
            pre[1][4].sync_counts()             # already landed: claim it before the next readback is queued
        keep = pre if (pre is not None and pre[0] is points) else None
        # explicit schedule (no autograd) when the model offers one and nothing non-standard was asked for
        explicit = (self.explicit_schedule and not kw and hasattr(self.model, "train_step_explicit")
                    and getattr(getattr(self.model, "backbone", None), "fused", False)
                    and getattr(self.model.voxel_encoder, "use_fused", True))
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        works = []

        def segment_ready(i):
            # called by the model when every gradient of segment i has been enqueued (with the stream current that
            # carries its last writer): start its all-reduce now, under the rest of the backward (RCCL runs it on its own
            # stream after the kernels enqueued so far on the current one)
            a, b, _ = self.flat.segments[i]
            works.append((i, dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=_grad_group(), async_op=True)))
        nseg = len(self.flat.segments)
        early = (lambda: segment_ready(0)) if (world > 1 and nseg > 1) else None
        enc_ready = (lambda: segment_ready(1)) if (world > 1 and nseg > 2) else None
        run = (lambda p: self.model.train_step_explicit(p, on_early_grads=early, next_points=next_points,
                                                        on_encoder_grads=enc_ready)) if explicit else \
            (lambda p: self.model.forward_train(p, None, **kw))
        if explicit:
            self.model._prefetched = keep
            losses = run(points)                 # enqueues the next batch's stage 1 itself (on its side stream)
        elif next_points is not None and hasattr(self.model, "prefetch"):
            self.model.prefetch(next_points)
            nxt = self.model._prefetched
            self.model._prefetched = keep
            losses = run(points)
            self.model._prefetched = nxt
        else:
            losses = run(points)
        if not explicit:
            vec = getattr(self.model, "_loss_vector", None)     # the fused path's [6] loss tensor: one sum, not five adds
            total = vec.sum() if vec is not None else sum(losses.values())
            self.model._loss_vector = None
            total.backward()
            self.flat.check_views()
        if self.fused_optimizer:
            if world > 1:                                                    # the 1/world rides in the update pass
                done = {i for i, _ in works}
                for i, (a, b, _) in enumerate(self.flat.segments):
                    if i not in done:
                        works.append((i, dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=_grad_group(), async_op=True)))
                _wait_all(works)
                tap = getattr(self, "on_reduced_grad", None)
                if tap is not None:                  # test tap: the summed gradient buffer the optimizer is about to read
                    tap(self.flat.grad)
            gnorm = self.opt.fused_clip_step(self.grad_clip.get("max_norm", 0.0), 1.0 / world, zero_grad=True)
            self._grads_clean = True
            packed = getattr(getattr(self.model, "backbone", None), "_packed", None)
            if explicit and packed is not None:
                # bf16 MFMA-layout copies of the updated weights for the next step, off its critical path
                # (on the decoder-B stream: idle until the next step's VFE forward is done, while the geometry stream's
                # chain of mask -> window layouts gates the next encoder and should start the moment the step does)
                from . import ops
                main, side = torch.cuda.current_stream(), ops.side_streams()["dec_b"]
                step_end = main.record_event()
                side.wait_event(step_end)
                with torch.cuda.stream(side):
                    packed.prepack()
                    packed.ready = side.record_event()
                # the next step orders its side streams behind this same event instead of recording two more on the
                # main stream (an event record is a packet of its own in the queue: ~5 us each between two steps)
                self.model._step_end_event = step_end
                self._prepack_flat_version = self.flat.flat._version
        else:
            allreduce_gradients(self.flat)
            gnorm = clip_grad_norm(self.flat, **self.grad_clip)
            self.opt.step()
        self.iter += 1
        return losses, gnorm

    # ---- checkpoint in the layout mmcv's CheckpointHook writes ({'meta', 'state_dict', 'optimizer'}), so that
    # epoch_N.pth interchanges with the reference's fine-tune config (load_from, configs/pre_sst/...:280)
    def save_checkpoint(self, path, meta=None):
        m = dict(iter=self.iter, epoch=0)
        m.update(meta or {})
        sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        torch.save(dict(meta=m, state_dict=sd, optimizer=self.opt.state_dict()), path)

    def load_checkpoint(self, path, strict=True):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["state_dict"], strict=strict)        # parameters are views of the flat buffer
        self.flat.check_storage()
        if ck.get("optimizer"):
            self.opt.load_state_dict(ck["optimizer"])
        if self.engine is not None:                      # the parameters and Adam's counter changed under the engine
            self.engine.invalidate_packed()
            self.engine.set_optimizer_steps(self.opt.step_count)
        self.iter = int(ck.get("meta", {}).get("iter", 0))
        if self.engine is not None:                      # a resumed run continues the uninterrupted run's mask stream
            self.engine.set_mask_draws(self.iter)
        return ck.get("meta", {})
