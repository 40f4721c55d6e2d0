"""Operator layer: the reference's `mmdet3d.ops` names for the pre-training hot path, backed by
libgeomae_hip (hand-written gfx950 kernels) through the C ABI in include/geomae_hip.h.

Mirrors (same names / argument meaning / error behaviour):
  dynamic_voxelize, Voxelization      mmdet3d/ops/voxel/voxelize.py:14-121 (+ voxel_layer pybind)
  scatter_v2                          mmdet3d/ops/sst/sst_ops.py:8-39
PyTorch is used for device memory and streams only; every function raises if the HIP library is
missing or a tensor is not a contiguous CUDA (ROCm) tensor -- there is no CPU fallback.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from ._lib import GeomaeTargetConfig, GeomaeWindowConfig, check, f3


# optional per-kernel timing (bench.py): name -> list of (start_event, end_event) recorded on the
# stream the kernel is launched on (torch's current stream == the stream handed to the C ABI)
KERNEL_EVENTS = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if KERNEL_EVENTS is not None and self.name in KERNEL_EVENTS:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if KERNEL_EVENTS is not None and self.name in KERNEL_EVENTS:
            self.e1.record()
            KERNEL_EVENTS[self.name].append((self.e0, self.e1))
        return False


# optional phase marks (tools/archive/phase_events.py): list of (name, event) recorded on the current stream at the phase
# boundaries of the explicit training schedule; None = disabled (one attribute test per mark)
PHASE_MARKS = None


def mark(name):
    if PHASE_MARKS is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        PHASE_MARKS.append((name, e))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Handle of torch's current HIP stream.  torch.cuda.current_stream() builds a Stream object through several
    Python layers (12 us per call, 24 calls per step); the raw accessor is a single C call."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _check_input(t, name, dtype=None):
    # same contract as the reference's CHECK_INPUT (voxelization_cuda.cu:8-14)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (geomae_amd has no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")


# ------------------------------------------------------------------------------------ A1
def grid_size(voxel_size, coors_range):
    """fp32 grid size (x, y, z) exactly as the reference computes it (voxelization_cuda.cu:375-377)."""
    out = (ctypes.c_int32 * 3)()
    check(_lib.load().geomae_grid_size(f3(voxel_size), f3(coors_range), out), "geomae_grid_size")
    return [int(v) for v in out]


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """Drop-in for voxel_layer.dynamic_voxelize: fills the pre-allocated coors [N,3] int32 (z,y,x)."""
    if NDim != 3:
        raise RuntimeError("only NDim == 3 is supported")
    _check_input(points, "points", torch.float32)
    _check_input(coors, "coors", torch.int32)
    if points.dim() != 2 or coors.shape != (points.shape[0], 3):
        raise RuntimeError("points must be [N, C] and coors [N, 3]")
    check(_lib.load().geomae_dynamic_voxelize(_ptr(points), points.shape[0], points.shape[1], f3(voxel_size),
                                              f3(coors_range), _ptr(coors), _stream()), "geomae_dynamic_voxelize")


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels, NDim=3):
    """Drop-in for voxel_layer.hard_voxelize (voxelization.h:58-76): fills the pre-allocated voxels
    [max_voxels, max_points, C], coors [max_voxels, 3] int32 (z,y,x), num_points_per_voxel [max_voxels] int32 and
    returns the number of voxels (one device->host readback, as the reference's return value implies)."""
    if NDim != 3:
        raise RuntimeError("only NDim == 3 is supported")
    _check_input(points, "points", torch.float32)
    _check_input(voxels, "voxels", torch.float32)
    _check_input(coors, "coors", torch.int32)
    _check_input(num_points_per_voxel, "num_points_per_voxel", torch.int32)
    n, C = points.shape
    if voxels.shape != (max_voxels, max_points, C) or coors.shape != (max_voxels, 3) or \
            num_points_per_voxel.shape != (max_voxels,):
        raise RuntimeError("voxels [max_voxels, max_points, C], coors [max_voxels, 3], num_points_per_voxel [max_voxels]")
    lib = _lib.load()
    wsb = lib.geomae_hard_voxelize_workspace_bytes(n, f3(voxel_size), f3(coors_range))
    if wsb < 0:
        raise RuntimeError("hard_voxelize: empty voxel grid")
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=points.device)
    num = torch.empty(1, dtype=torch.int32, device=points.device)
    check(lib.geomae_hard_voxelize(_ptr(points), n, C, f3(voxel_size), f3(coors_range), int(max_points), int(max_voxels),
                                   _ptr(voxels), _ptr(coors), _ptr(num_points_per_voxel), _ptr(num), _ptr(ws), wsb,
                                   _stream()), "geomae_hard_voxelize")
    return int(num.item())


class Voxelization(nn.Module):
    """mmdet3d.ops.Voxelization (voxelize.py:63-121): dynamic mode (max_num_points == -1 or max_voxels == -1) ->
    coors [N,3]; hard mode -> (voxels [M, max_points, C], coors [M,3], num_points_per_voxel [M])."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else (max_voxels, max_voxels)
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid
        self.pcd_shape = [*grid[:2].tolist(), 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        if self.max_num_points == -1 or max_voxels == -1:
            coors = input.new_zeros(size=(input.size(0), 3), dtype=torch.int)
            dynamic_voxelize(input.contiguous(), coors, self.voxel_size, self.point_cloud_range, 3)
            return coors
        input = input.contiguous()
        voxels = input.new_zeros(size=(max_voxels, self.max_num_points, input.size(1)))
        coors = input.new_zeros(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = input.new_zeros(size=(max_voxels,), dtype=torch.int)
        voxel_num = hard_voxelize(input, voxels, coors, num_points_per_voxel, self.voxel_size, self.point_cloud_range,
                                  self.max_num_points, max_voxels, 3)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range="
                f"{self.point_cloud_range}, max_num_points={self.max_num_points}, max_voxels={self.max_voxels})")


class Voxelization_with_flag(Voxelization):
    """mmdet3d.ops.Voxelization_with_flag (ops/voxel/voxelize.py:126-244): hard voxelization that also returns
    voxels_flag [M, max_points] bool, True for the slots that hold a point (assign_flag_to_voxel,
    voxelization_cuda.cu:110-127, sets the slot of every kept point).  The mae_sst config constructs two of these
    (hard_sub_voxel_layer_low / _med) and never calls them."""

    def forward(self, input):
        out = super().forward(input)
        if not isinstance(out, tuple):
            return out                                   # dynamic mode: coors only, as the reference
        voxels, coors, num = out
        slots = torch.arange(self.max_num_points, device=num.device, dtype=num.dtype)
        return voxels, slots[None, :] < num[:, None], coors, num


def voxelize_batch3(points, batch_offsets, batch_size, vs_top, vs_med, vs_low, coors_range):
    """Three resolutions in one pass over the concatenated batch -> three [N,4] int32 (b,z,y,x)."""
    _check_input(points, "points", torch.float32)
    _check_input(batch_offsets, "batch_offsets", torch.int32)
    n = points.shape[0]
    out = [torch.empty((n, 4), dtype=torch.int32, device=points.device) for _ in range(3)]
    check(_lib.load().geomae_voxelize_batch3(_ptr(points), n, points.shape[1], _ptr(batch_offsets), batch_size,
                                             f3(vs_top), f3(vs_med), f3(vs_low), f3(coors_range), _ptr(out[0]),
                                             _ptr(out[1]), _ptr(out[2]), _stream()), "geomae_voxelize_batch3")
    return out


# ------------------------------------------------------------------------------------ A2
class ZeroArena:
    """One zero-filled device buffer carved into a training step's accumulator / atomics-target buffers (BatchNorm sums,
    pillar-mean sums, max-pool outputs, gradient rows that only some kernels write ...).  The library zeroes each such
    buffer with its own hipMemsetAsync -- a ~5 us fill kernel apiece, a dozen of them between DEPENDENT kernels of a
    step.  The explicit schedule instead fills one arena on a side stream and calls the entry points under
    `prezeroed()`, which tells the library (per host thread) to skip its memsets."""

    def __init__(self, nbytes, device):
        self.buf = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        self.off = 0

    @staticmethod
    def nbytes(*specs):
        """specs: (shape, dtype) pairs -> bytes needed (each carve is 256-byte aligned)."""
        total = 0
        for shape, dtype in specs:
            n = torch.empty((), dtype=dtype).element_size()
            for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
                n *= int(d)
            total += (n + 255) // 256 * 256
        return total

    def take(self, shape, dtype):
        shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        n = torch.empty((), dtype=dtype).element_size()
        for d in shape:
            n *= int(d)
        start = self.off
        self.off += (n + 255) // 256 * 256
        if self.off > self.buf.numel():
            raise RuntimeError("ZeroArena: carved past the end (size the arena with ZeroArena.nbytes)")
        return self.buf[start:start + n].view(dtype).view(shape)


class prezeroed:
    """with prezeroed(): ...  -- entry points called inside trust that their accumulator outputs are already zero
    (geomae_set_accumulators_prezeroed, include/geomae_hip.h lists which ones honour it)."""

    def __enter__(self):
        check(_lib.load().geomae_set_accumulators_prezeroed(1), "geomae_set_accumulators_prezeroed")
        return self

    def __exit__(self, *a):
        check(_lib.load().geomae_set_accumulators_prezeroed(0), "geomae_set_accumulators_prezeroed")
        return False


def _zeros_or_empty(zeros, shape, dtype, device):
    return zeros.take(shape, dtype) if zeros is not None else torch.empty(shape, dtype=dtype, device=device)


_SIDE_STREAMS = {}
STREAM_PROBE = {}          # device key -> what the probe saw (bench.py reports it)


def _spin_us(dev, target_us=500.0):
    """cycles of torch.cuda._sleep that last ~target_us on this device.  Calibrated after a warm-up (at idle clocks the
    same cycle count lasts twice as long as a moment later) and long against the ~40 us of launch + synchronise that the
    host-side timing of wait_blocks contains: with 120 us spins calibrated cold, a blocked victim (2 T + o) measured
    1.6 x the free one (T + o) -- exactly the threshold -- and a run picked side streams on the main stream's queue."""
    cyc = 400_000
    for _ in range(8):                                         # ~ms of work: clocks up
        torch.cuda._sleep(cyc)
    for _ in range(3):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(cyc)
        e1.record()
        e1.synchronize()
        us = max(e0.elapsed_time(e1) * 1e3, 1.0)
        cyc = max(10_000, int(cyc * target_us / us))
    return cyc


def wait_blocks(victim, waiter, helper, cycles, dev):
    """Does a cross-stream WAIT on `waiter` hold up a kernel on `victim`?  (-> blocked, seconds)

    Kernels of two streams that share a hardware queue may still run side by side (their packets carry no barrier
    bit), so a pair of spin kernels does not tell whether two streams share one.  What a shared queue cannot do is let
    one stream wait while the other goes on: a cross-stream wait is a barrier packet, and everything behind it in the
    hardware queue -- whichever stream it came from -- stays behind it.  (That is how a step's two decoder stacks ended
    up running one after the other once the geometry stream, which waits for the decoder-B stream, shared the main
    stream's queue.)  Here `helper` spins for ~500 us, `waiter` waits for it, and `victim` runs a spin of the same
    length: alone it takes T, behind the wait 2 T.  The helper's own event record is a barrier packet too, so a
    victim that shares the HELPER's queue is also reported -- all three streams of a schedule are meant to sit on
    different queues anyway (_pick_side_streams tries every assignment of the roles)."""
    import time
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(helper):
        torch.cuda._sleep(cycles)
        ev = helper.record_event()
    waiter.wait_event(ev)
    with torch.cuda.stream(waiter):
        torch.cuda._sleep(100)
    t0 = time.perf_counter()
    with torch.cuda.stream(victim):
        torch.cuda._sleep(cycles)
    victim.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    return dt, t0


def comm_blocks(victim, issuer, helper, group, cycles, dev):
    """Does a collective of `group` issued from `issuer` hold up a kernel on `victim`?  A stream-ordered collective makes
    the communicator's own stream WAIT for the issuing stream; if that stream shares the victim's hardware queue the
    victim stays behind the wait (wait_blocks with the communicator's stream as the waiter: the step's hooks issue
    collectives from the side streams, and the main stream must not stand behind them).  -> seconds of the victim's spin"""
    import time
    from torch import distributed as dist
    t = torch.zeros(8, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(helper):
        torch.cuda._sleep(cycles)
        ev = helper.record_event()
    issuer.wait_event(ev)
    with torch.cuda.stream(issuer):
        work = dist.all_reduce(t, group=group, async_op=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(victim):
        torch.cuda._sleep(cycles)
    victim.synchronize()
    dt = time.perf_counter() - t0
    with torch.cuda.stream(issuer):
        work.wait()
    torch.cuda.synchronize(dev)
    return dt


def _spin_seconds(stream, cycles, dev):
    import time
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cycles)
    stream.synchronize()
    return time.perf_counter() - t0


def streams_independent(triple, cycles, dev):
    """All three streams on different hardware queues: no assignment of (victim, waiter, helper) delays the victim."""
    import itertools
    alone = min(_spin_seconds(s, cycles, dev) for s in triple for _ in range(2))
    for v, w, h in itertools.permutations(triple):
        dt = min(wait_blocks(v, w, h, cycles, dev)[0] for _ in range(2))
        if dt > 1.5 * alone:                                   # (blocked: ~1.9 x with 500 us spins)
            return False
    return True


def _pick_side_streams(dev, names):
    """Streams for `names` (two of them) that share a hardware queue neither with the current (main) stream nor with
    each other.

    ROCm multiplexes a process's HIP streams onto a few hardware queues per priority level (4 by default); which queue a
    stream lands on depends on what else created streams before it -- torch's stream pool, RCCL's communicators and
    their internal streams.  With the geometry stream on the main stream's queue the two decoder stacks of a step ran
    one after the other (2.60 instead of 2.11 ms per step, profiles/r03_fx_timeline_before.txt: what the world > 1
    schedule looked like once RCCL's streams existed).  Creation order cannot fix that in general, so the mapping is
    MEASURED: pairs of candidates from torch's pool are probed together with main (streams_independent) and the first
    pair that passes wins."""
    import os
    assert len(names) == 2
    main = torch.cuda.current_stream(dev)
    key = (dev.type, dev.index)
    if os.environ.get("GEOMAE_STREAM_PROBE", "1") == "0":
        STREAM_PROBE[key] = dict(probed=False)
        return {n: torch.cuda.Stream(device=dev) for n in names}
    cycles = _spin_us(dev)
    cands = []
    for _ in range(8):
        c = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(c):
            torch.zeros(1, device=dev)                         # first use = queue assignment
        cands.append(c)
    torch.cuda.synchronize(dev)
    chosen, tried = None, 0
    for j in range(1, len(cands)):
        for i in range(j):
            tried += 1
            if streams_independent((main, cands[i], cands[j]), cycles, dev):
                chosen = (cands[i], cands[j])
                break
        if chosen:
            break
    STREAM_PROBE[key] = dict(probed=True, pairs_tried=tried, distinct_queues=chosen is not None, spin_cycles=cycles)
    if chosen is None:                                         # fewer queues than streams: any two
        chosen = (cands[0], cands[1])
    return dict(zip(names, chosen))


def side_streams(device=None):
    """The step's side streams {"geo", "dec_b"}, one set per device (+ "dec_a" / "prefetch" on request: extra_stream).

    The explicit training schedule uses main + these TWO streams and needs each on a hardware queue of its own; they are
    chosen by measurement (_pick_side_streams) at first use -- at world > 1 the trainer calls this AFTER its process
    groups have created their communicators and streams (Trainer.__init__), so that the probe sees the final picture.
    reset_side_streams() forgets the choice (a process group created later may have re-dealt the queues)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        import os
        # GEOMAE_SIDE_STREAMS=3: also "dec_a" up front.  Only for the test hook that runs several ranks on ONE GPU with
        # gloo (bench.py GEOMAE_BENCH_SHARE_GPU, tests/test_gpu_multirank.py): two processes then own 8 queues on one
        # device, and gloo's copy stream alone on a queue of its own waited ~100 ms per collective for a time slice.
        three = os.environ.get("GEOMAE_SIDE_STREAMS") == "3"
        if three:
            st = {}
            for name in ("geo", "dec_a", "dec_b"):
                st[name] = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st[name]):
                    torch.zeros(1, device=dev)
            STREAM_PROBE[key] = dict(probed=False)
        else:
            st = _pick_side_streams(dev, ("geo", "dec_b"))
        torch.cuda.synchronize(dev)
        _SIDE_STREAMS[key] = st
    return _SIDE_STREAMS[key]


def reset_side_streams(device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    _SIDE_STREAMS.pop((dev.type, dev.index), None)


def extra_stream(name, device=None):
    """A further stream of the set ("dec_a", "prefetch"), created on first request (after the fixed ones)."""
    st = side_streams(device)
    if name not in st:
        st[name] = torch.cuda.Stream(device=st["geo"].device)
    return st[name]


def prefetch_stream(device=None):
    return extra_stream("prefetch", device)


class PillarSegments:
    """Result of the one counting sort per batch that replaces the reference's six unique(dim=0)."""
    __slots__ = ("cell_table", "voxel_coors", "inv", "order", "seg_start", "sample_start", "num_pillars",
                 "cap", "grid", "batch_size", "num_points", "_host", "_pinned", "_event")

    def start_readback(self):
        """Queue the iteration's one device->host readback (pillar offsets per sample) without waiting: an
        async copy into pinned memory + an event.  A caller that prepares batch k+1 while step k is still
        being enqueued (detector.prefetch) finds the counts on the host when it needs them and never stalls."""
        if self._host is None and self._event is None:
            self._pinned = torch.empty(self.sample_start.shape, dtype=torch.int32, pin_memory=True)
            self._pinned.copy_(self.sample_start, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()

    def sync_counts(self):
        """Pillar offsets per sample on the host (waits for the readback if it has not landed yet)."""
        if self._host is None:
            self.start_readback()
            self._event.synchronize()
            self._host = self._pinned.tolist()
        return self._host

    @property
    def V(self):
        return self.sync_counts()[-1]


def pillar_segment(coors, batch_size, grid_zyx, cap=None):
    _check_input(coors, "coors", torch.int32)
    n = coors.shape[0]
    gz, gy, gx = [int(g) for g in grid_zyx]
    cells = batch_size * gz * gy * gx
    cap = min(n, cells) if cap is None else cap
    dev = coors.device
    lib = _lib.load()
    s = PillarSegments()
    s.cell_table = torch.empty(cells, dtype=torch.int32, device=dev)
    s.voxel_coors = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=dev)
    s.inv = torch.empty(n, dtype=torch.int32, device=dev)
    s.order = torch.empty(n, dtype=torch.int32, device=dev)
    s.seg_start = torch.empty(max(cap, 1) + 1, dtype=torch.int32, device=dev)
    s.sample_start = torch.empty(batch_size + 1, dtype=torch.int32, device=dev)
    s.num_pillars = torch.empty(1, dtype=torch.int32, device=dev)
    s.cap, s.grid, s.batch_size, s.num_points, s._host = cap, (gz, gy, gx), batch_size, n, None
    s._pinned = s._event = None
    wsb = lib.geomae_pillar_segment_workspace_bytes(n, batch_size, gz, gy, gx)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    check(lib.geomae_pillar_segment(_ptr(coors), n, batch_size, gz, gy, gx, _ptr(s.cell_table), _ptr(s.voxel_coors),
                                    _ptr(s.inv), _ptr(s.order), _ptr(s.seg_start), _ptr(s.sample_start),
                                    _ptr(s.num_pillars), _ptr(ws), wsb, _stream()), "geomae_pillar_segment")
    return s


def pillar_segment_nd(coors, batch_size, grid_zyx):
    """pillar_segment for coors [N, 3] (z,y,x; batch_size 1) or [N, 4]: rows outside the grid are dropped (inv = -1)."""
    _check_input(coors, "coors", torch.int32)
    n, ndim = coors.shape
    gz, gy, gx = [int(g) for g in grid_zyx]
    cells = batch_size * gz * gy * gx
    cap = max(min(n, cells), 1)
    dev, lib = coors.device, _lib.load()
    s = PillarSegments()
    s.cell_table = torch.empty(cells, dtype=torch.int32, device=dev)
    s.voxel_coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)        # always (b, z, y, x) rows
    s.inv = torch.empty(n, dtype=torch.int32, device=dev)
    s.order = torch.empty(n, dtype=torch.int32, device=dev)
    s.seg_start = torch.empty(cap + 1, dtype=torch.int32, device=dev)
    s.sample_start = torch.empty(batch_size + 1, dtype=torch.int32, device=dev)
    s.num_pillars = torch.empty(1, dtype=torch.int32, device=dev)
    s.cap, s.grid, s.batch_size, s.num_points, s._host = cap, (gz, gy, gx), batch_size, n, None
    s._pinned = s._event = None
    wsb = lib.geomae_pillar_segment_workspace_bytes(n, batch_size, gz, gy, gx)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    check(lib.geomae_pillar_segment_nd(_ptr(coors), ndim, n, batch_size, gz, gy, gx, _ptr(s.cell_table), _ptr(s.voxel_coors),
                                       _ptr(s.inv), _ptr(s.order), _ptr(s.seg_start), _ptr(s.sample_start),
                                       _ptr(s.num_pillars), _ptr(ws), wsb, _stream()), "geomae_pillar_segment_nd")
    return s


def segment_mean_xyz(points, seg, zeros=None):
    """Pillar means of the xyz columns from the pillar-sorted point list (no atomics, no workspace; `zeros` is unused and
    kept for callers of the former atomic version)."""
    mean = torch.empty((max(seg.cap, 1), 3), dtype=torch.float32, device=points.device)
    check(_lib.load().geomae_segment_mean_xyz_sorted(_ptr(points), points.shape[1], _ptr(seg.order), _ptr(seg.seg_start),
                                                     _ptr(seg.num_pillars), seg.cap, _ptr(mean), _stream()),
          "geomae_segment_mean_xyz_sorted")
    return mean


class _SegmentMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, seg, V):
        feat = feat.contiguous()
        _check_input(feat, "feat", torch.float32)
        C = feat.shape[1]
        out = torch.empty((V, C), dtype=torch.float32, device=feat.device)
        arg = torch.empty((V, C), dtype=torch.int32, device=feat.device)
        check(_lib.load().geomae_segment_max_forward(_ptr(feat), C, _ptr(seg.order), _ptr(seg.seg_start),
                                                     _ptr(seg.num_pillars), V, _ptr(out), _ptr(arg), _stream()),
              "geomae_segment_max_forward")
        ctx.seg, ctx.n, ctx.C = seg, feat.shape[0], C
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        grad_out = grad_out.contiguous().float()
        grad = torch.empty((ctx.n, ctx.C), dtype=torch.float32, device=grad_out.device)
        check(_lib.load().geomae_segment_max_backward(_ptr(grad_out), _ptr(arg), _ptr(ctx.seg.inv), ctx.n, ctx.C,
                                                      _ptr(grad), _stream()), "geomae_segment_max_backward")
        return grad, None, None


def segment_max(feat, seg, V=None):
    return _SegmentMax.apply(feat, seg, seg.V if V is None else V)


def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None, grid_zyx=None,
               batch_size=None):
    """mmdet3d.ops.scatter_v2 (sst_ops.py:8-39) on the pillar-segment kernels.

    `coors` is [N,4] int32 (b,z,y,x).  The dense cell table needs the grid extent: pass grid_zyx /
    batch_size, or they are taken from coors.max() (one extra sync -- fine for the operator API;
    the fused detector path never calls this wrapper)."""
    assert feat.size(0) == coors.size(0)
    if mode == "avg":
        mode = "mean"
    if min_points > 0:
        raise NotImplementedError("min_points > 0 is not on the pre-training path")
    if isinstance(unq_inv, PillarSegments):
        seg = unq_inv
    else:
        coors = coors.contiguous().int()
        if grid_zyx is None:
            mx = coors.max(0).values.tolist()
            batch_size, grid_zyx = mx[0] + 1, (mx[1] + 1, mx[2] + 1, mx[3] + 1)
        seg = pillar_segment(coors, batch_size, grid_zyx)
    V = seg.V
    if mode == "max":
        new_feat = segment_max(feat.float(), seg, V)
    elif mode == "mean":
        if feat.shape[1] != 3:
            raise NotImplementedError("segment mean is implemented for the xyz cluster centre (3 channels)")
        new_feat = segment_mean_xyz(feat.contiguous().float(), seg)[:V]
    else:
        raise NotImplementedError(mode)
    out_coors = seg.voxel_coors[:V]
    if not return_inv:
        return new_feat, out_coors
    return new_feat, out_coors, seg.inv.long()


# ------------------------------------------------------------------------------------ N3 DynamicScatter
_REDUCE = dict(sum=0, mean=1, max=2)


def dynamic_point_to_voxel_forward(feats, coors, reduce_type="max", grid_zyx=None, batch_size=None):
    """Drop-in for voxel_layer.dynamic_point_to_voxel_forward (voxelization.h:112-134): -> [reduced_feats [M,C],
    out_coors [M,ndim] (lexicographic), coors_map [N] int32 (-1 = dropped row), reduce_count [M] int32].
    The dense-grid counting sort needs the coordinate bounds: pass grid_zyx (DynamicScatter does), or they are
    read from coors.max() (one more host sync; the reference's unique_dim syncs for the output size anyway)."""
    _check_input(feats, "feats", torch.float32)
    _check_input(coors, "coors", torch.int32)
    n, C = feats.shape
    ndim = coors.shape[1]
    if n == 0:      # scatter_points_cuda.cu:193-197
        e = torch.empty(0, dtype=torch.int32, device=feats.device)
        return [feats.clone().detach(), coors.clone().detach(), e, e.clone()]
    if ndim not in (3, 4):
        raise RuntimeError("coors must be [N,3] (z,y,x) or [N,4] (b,z,y,x)")
    if grid_zyx is None or (ndim == 4 and batch_size is None):
        mx = [max(int(v), 0) + 1 for v in coors.max(0).values.tolist()]
        if grid_zyx is None:
            grid_zyx = mx[-3:]
        if ndim == 4 and batch_size is None:
            batch_size = mx[0]
    nb = int(batch_size) if ndim == 4 else 1
    gz, gy, gx = [int(g) for g in grid_zyx]
    cap = min(n, nb * gz * gy * gx)
    dev, lib = feats.device, _lib.load()
    reduced = torch.empty((cap, C), dtype=torch.float32, device=dev)
    out_coors = torch.empty((cap, ndim), dtype=torch.int32, device=dev)
    cmap = torch.empty(n, dtype=torch.int32, device=dev)
    count = torch.empty(cap, dtype=torch.int32, device=dev)
    num = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = lib.geomae_dynamic_point_to_voxel_workspace_bytes(n, cap, nb, gz, gy, gx)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    check(lib.geomae_dynamic_point_to_voxel_forward(_ptr(feats), _ptr(coors), n, C, ndim, nb, gz, gy, gx,
                                                    _REDUCE[reduce_type], cap, _ptr(reduced), _ptr(out_coors), _ptr(cmap),
                                                    _ptr(count), _ptr(num), _ptr(ws), wsb, _stream()),
          "geomae_dynamic_point_to_voxel_forward")
    M = int(num.item())
    return [reduced[:M], out_coors[:M], cmap, count[:M]]


def dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats, coors_map, reduce_count,
                                    reduce_type="max"):
    """Drop-in for voxel_layer.dynamic_point_to_voxel_backward (voxelization.h:136-154): fills grad_feats [N,C]."""
    n, C = feats.shape
    M = reduced_feats.shape[0]
    ws = torch.empty(max(M * C, 1), dtype=torch.int32, device=feats.device) if reduce_type == "max" else None
    check(_lib.load().geomae_dynamic_point_to_voxel_backward(
        _ptr(grad_feats), _ptr(grad_reduced_feats.contiguous()), _ptr(feats), _ptr(reduced_feats), _ptr(coors_map),
        _ptr(reduce_count), n, M, C, _REDUCE[reduce_type], _ptr(ws), _stream()), "geomae_dynamic_point_to_voxel_backward")


class _DynamicScatterFn(torch.autograd.Function):
    """mmdet3d/ops/voxel/scatter_points.py:9-47 (_dynamic_scatter)."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type, grid_zyx, batch_size):
        voxel_feats, voxel_coors, p2v, cnt = dynamic_point_to_voxel_forward(feats, coors, reduce_type, grid_zyx, batch_size)
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, voxel_feats, p2v, cnt)
        ctx.mark_non_differentiable(voxel_coors)
        return voxel_feats, voxel_coors

    @staticmethod
    def backward(ctx, grad_voxel_feats, grad_voxel_coors=None):
        feats, voxel_feats, p2v, cnt = ctx.saved_tensors
        grad_feats = torch.zeros_like(feats)
        if feats.shape[0] > 0:
            dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats.contiguous().float(), feats, voxel_feats, p2v, cnt,
                                            ctx.reduce_type)
        return grad_feats, None, None, None, None


def dynamic_scatter(feats, coors, reduce_type="max", grid_zyx=None, batch_size=None):
    return _DynamicScatterFn.apply(feats, coors, reduce_type, grid_zyx, batch_size)


class DynamicScatter(nn.Module):
    """mmdet3d.ops.DynamicScatter (ops/voxel/scatter_points.py:53-99): scatters point features into voxels by mean
    (average_points=True) or max.  coors [N,3] (z,y,x) or [N,4] (b,z,y,x); the reference loops over the samples of a
    batch and concatenates -- the same rows in the same order as one lexicographic (b,z,y,x) grouping."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size, self.point_cloud_range, self.average_points = voxel_size, point_cloud_range, average_points
        # The reference op groups ANY non-negative coordinates (its own test feeds z up to 19 to a module whose
        # range holds one z cell), so the table extent comes from coors.max() (one readback; unique_dim in the
        # reference syncs for its output size too).  Set `grid_zyx` to skip it when the caller knows the bounds.
        self.grid_zyx = None

    def forward_single(self, points, coors):
        return dynamic_scatter(points.contiguous(), coors.contiguous(), "mean" if self.average_points else "max", self.grid_zyx)

    def forward(self, points, coors, batch_size=None):
        if coors.size(-1) == 3 or coors.shape[0] == 0:
            return self.forward_single(points, coors)
        if batch_size is None:
            batch_size = int(coors[-1, 0].item()) + 1            # as the reference (scatter_points.py:81)
        return dynamic_scatter(points.contiguous(), coors.contiguous(), "mean" if self.average_points else "max",
                               self.grid_zyx, batch_size)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range={self.point_cloud_range}, "
                f"average_points={self.average_points})")


# ------------------------------------------------------------------------------------ A6
def random_mask_launch(seg, keep_fraction, seed, window_cfg=None):
    """Enqueue the mask kernel without needing the pillar counts on the host (buffers sized by the segment capacity):
    a training loop runs it for the NEXT batch together with that batch's voxelization.  -> raw state for
    random_mask_finish.  window_cfg (GeomaeWindowConfig): the same subset, ids emitted window-major
    (geomae_random_mask_windowed: the token order the SST stacks want)."""
    dev = seg.voxel_coors.device
    cap = max(int(seg.cap), 1)
    ids_keep = torch.empty(cap, dtype=torch.int32, device=dev)
    ids_mask = torch.empty(cap, dtype=torch.int32, device=dev)
    token_row = torch.empty(cap, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    if window_cfg is not None:
        check(_lib.load().geomae_random_mask_windowed(_ptr(seg.sample_start), seg.batch_size, float(keep_fraction),
                                                      int(seed) & (2 ** 64 - 1), _ptr(seg.voxel_coors), ctypes.byref(window_cfg),
                                                      _ptr(ids_keep), _ptr(ids_mask), _ptr(token_row), _ptr(counts), _stream()),
              "geomae_random_mask_windowed")
    else:
        check(_lib.load().geomae_random_mask(_ptr(seg.sample_start), seg.batch_size, float(keep_fraction),
                                             int(seed) & (2 ** 64 - 1), _ptr(ids_keep), _ptr(ids_mask), _ptr(token_row),
                                             _ptr(counts), _stream()), "geomae_random_mask")
    return ids_keep, ids_mask, token_row, counts, float(keep_fraction)


def random_mask_finish(raw, seg):
    """Slice the mask buffers with the host-side pillar counts (per sample int(L * keep_fraction) kept, as the kernel)."""
    ids_keep, ids_mask, token_row, counts, keep_fraction = raw
    starts = seg.sync_counts()
    V = starts[-1]
    n_keep = sum(int((starts[b + 1] - starts[b]) * keep_fraction) for b in range(seg.batch_size))
    return ids_keep[:n_keep], ids_mask[:V - n_keep], token_row[:V], counts


def random_mask(seg, keep_fraction, seed, window_cfg=None):
    """-> ids_keep [n_keep], ids_mask [n_mask] (int32; ascending, or window-major with window_cfg), token_row [V] int32,
    counts [2] int32."""
    return random_mask_finish(random_mask_launch(seg, keep_fraction, seed, window_cfg), seg)


def gather_token_coors(ids_keep, ids_mask, voxel_coors):
    """-> coors [n_keep + n_mask, 4] int32 (kept rows, then masked rows), ids_keep as int64."""
    _check_input(ids_keep, "ids_keep", torch.int32)
    _check_input(ids_mask, "ids_mask", torch.int32)
    _check_input(voxel_coors, "voxel_coors", torch.int32)
    nk, nm = ids_keep.numel(), ids_mask.numel()
    out = torch.empty((nk + nm, 4), dtype=torch.int32, device=voxel_coors.device)
    ik64 = torch.empty(nk, dtype=torch.int64, device=voxel_coors.device)
    check(_lib.load().geomae_gather_token_coors(_ptr(ids_keep), nk, _ptr(ids_mask), nm, _ptr(voxel_coors), _ptr(out),
                                                _ptr(ik64), _stream()), "geomae_gather_token_coors")
    return out, ik64


def token_rows_from_ids(ids_keep, ids_mask, V):
    """token_row / counts for externally supplied ids (parity tests inject the reference's ids)."""
    dev = ids_keep.device
    token_row = torch.empty(V, dtype=torch.int32, device=dev)
    token_row[ids_keep.long()] = torch.arange(ids_keep.numel(), dtype=torch.int32, device=dev)
    token_row[ids_mask.long()] = torch.arange(ids_mask.numel(), dtype=torch.int32, device=dev) + ids_keep.numel()
    counts = torch.tensor([ids_keep.numel(), ids_mask.numel()], dtype=torch.int32, device=dev)
    return token_row, counts


# ------------------------------------------------------------------------------------ A5, A7-A11
def make_target_config(grid_size_zyx, ratio_low, ratio_med, vs_top, vs_med, vs_low, coors_range):
    c = GeomaeTargetConfig()
    c.grid_size[:] = [int(v) for v in grid_size_zyx]
    c.ratio_low[:] = [int(v) for v in ratio_low]
    c.ratio_med[:] = [int(v) for v in ratio_med]
    c.voxel_size_top[:] = [float(v) for v in vs_top]
    c.voxel_size_med[:] = [float(v) for v in vs_med]
    c.voxel_size_low[:] = [float(v) for v in vs_low]
    c.coors_range[:] = [float(v) for v in coors_range]
    return c


def geometry_targets(points, seg, coors_med, coors_low, cfg, token_row=None, counts=None, n_rows=None,
                     want_cov=False):
    """All geometric targets of extract_feat (ssl.py:185-219) in two kernels.  Rows = masked pillars in
    ids_mask order when token_row/counts are given (n_rows = n_mask), else one row per pillar."""
    dev = points.device
    V = seg.V
    M = V if token_row is None else n_rows
    s_low = cfg.ratio_low[0] * cfg.ratio_low[1] * cfg.ratio_low[2]
    s_med = cfg.ratio_med[0] * cfg.ratio_med[1] * cfg.ratio_med[2]
    f32, u8 = torch.float32, torch.uint8
    out = dict(
        centroid_low=torch.empty((M, s_low, 3), dtype=f32, device=dev),
        mask_low=torch.empty((M, s_low), dtype=u8, device=dev),
        centroid_med=torch.empty((M, s_med, 3), dtype=f32, device=dev),
        mask_med=torch.empty((M, s_med), dtype=u8, device=dev),
        centroid_top=torch.empty((M, 3), dtype=f32, device=dev),
        normal=torch.empty((M, 3), dtype=f32, device=dev),
        curv=torch.empty((M, 3), dtype=torch.float64, device=dev),
        top_raw=torch.empty((max(V, 1), 3), dtype=f32, device=dev),
        med_raw=torch.empty((max(V, 1), s_med, 3), dtype=f32, device=dev),
        med_raw_mask=torch.empty((max(V, 1), s_med), dtype=u8, device=dev),
        # the scatter matrices always go through memory: with this buffer the library runs the eigen-decompositions one
        # per thread in a second launch (targets.hip normal_eig_kernel) instead of on lane 0 of each pillar's wave
        cov=torch.empty((max(M, 1), 6), dtype=f32, device=dev),
        occ_counts=torch.empty(2, dtype=torch.int32, device=dev))
    check(_lib.load().geomae_geometry_targets(
        _ptr(points), points.shape[1], _ptr(seg.order), _ptr(seg.seg_start), _ptr(seg.num_pillars), V,
        _ptr(seg.voxel_coors), _ptr(coors_med), _ptr(coors_low), _ptr(seg.cell_table), seg.batch_size,
        _ptr(token_row), _ptr(counts), ctypes.byref(cfg), _ptr(out["centroid_low"]), _ptr(out["mask_low"]),
        _ptr(out["centroid_med"]), _ptr(out["mask_med"]), _ptr(out["centroid_top"]), _ptr(out["normal"]),
        _ptr(out["curv"]), _ptr(out["top_raw"]), _ptr(out["med_raw"]), _ptr(out["med_raw_mask"]), _ptr(out["cov"]),
        _ptr(out["occ_counts"]), M, _stream()), "geomae_geometry_targets")
    out["mask_low_u8"], out["mask_med_u8"] = out["mask_low"], out["mask_med"]
    out["mask_low"] = out["mask_low"].view(torch.bool)
    out["mask_med"] = out["mask_med"].view(torch.bool)
    return out


# ------------------------------------------------------------------------------------ A12-A19
def make_window_config(window_shape, shift, bev_shape):
    c = GeomaeWindowConfig()
    c.window_shape[:] = [int(v) for v in window_shape]
    c.shift[:] = [int(v) for v in shift]
    c.bev_shape[:] = [int(v) for v in bev_shape]
    return c


class WindowLayout:
    """CSR grouping of tokens by window for one shift (replaces the flat2win index dictionaries)."""
    __slots__ = ("win_start", "win_tokens", "tok_win", "tok_pos", "num_windows", "max_windows", "n", "max_tokens",
                 "bun_start", "num_bundles", "bun_tok", "pos_info", "fbun_tok", "num_fbundles", "fitems", "num_fitems")


def _new_window_layout(n, batch_size, wcfg, dev):
    nwx = (wcfg.bev_shape[0] + wcfg.window_shape[0] - 1) // wcfg.window_shape[0] + 1
    nwy = (wcfg.bev_shape[1] + wcfg.window_shape[1] - 1) // wcfg.window_shape[1] + 1
    slots = batch_size * nwx * nwy
    L = WindowLayout()
    L.n = n
    L.max_windows = max(1, min(n, slots))
    L.max_tokens = wcfg.window_shape[0] * wcfg.window_shape[1]
    L.win_start = torch.empty(L.max_windows + 1, dtype=torch.int32, device=dev)
    L.win_tokens = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    L.tok_win = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    L.tok_pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    L.num_windows = torch.empty(1, dtype=torch.int32, device=dev)
    L.bun_start = torch.empty(L.max_windows + 1, dtype=torch.int32, device=dev)
    L.num_bundles = torch.empty(1, dtype=torch.int32, device=dev)
    L.bun_tok = L.pos_info = None          # the attention plan: window_build_batch fills it
    L.fbun_tok = L.num_fbundles = None     # ... and the second packing (bundles of the one-launch layer kernel)
    L.fitems = L.num_fitems = None         # ... and the one-launch forward's work items (query-split bundles)
    return L


def window_build_batch(jobs, batch_size, wcfg):
    """jobs: list of (coors [n,4] int32, shift_index), at most 4 -> list of WindowLayout.  One launch per build stage
    for all jobs (geomae_window_build_batch): the serial chain is as long as for one layout."""
    from ._lib import GeomaeWindowBuildJob
    lib = _lib.load()
    arr = (GeomaeWindowBuildJob * len(jobs))()
    ns = (ctypes.c_int32 * len(jobs))()
    layouts, keep = [], []
    for k, (coors, shift_index) in enumerate(jobs):
        coors = coors.contiguous()
        _check_input(coors, "coors", torch.int32)
        keep.append(coors)
        L = _new_window_layout(coors.shape[0], batch_size, wcfg, coors.device)
        layouts.append(L)
        j = arr[k]
        j.coors, j.num_tokens, j.shift_index = coors.data_ptr(), L.n, shift_index
        j.win_start, j.win_tokens, j.tok_win = L.win_start.data_ptr(), L.win_tokens.data_ptr(), L.tok_win.data_ptr()
        j.tok_pos, j.num_windows = L.tok_pos.data_ptr(), L.num_windows.data_ptr()
        j.bun_start, j.num_bundles = L.bun_start.data_ptr(), L.num_bundles.data_ptr()
        L.bun_tok = torch.empty(L.max_windows + 1, dtype=torch.int32, device=coors.device)
        L.pos_info = torch.empty((max(L.n, 1), 4), dtype=torch.int32, device=coors.device)
        j.bun_tok, j.pos_info = L.bun_tok.data_ptr(), L.pos_info.data_ptr()
        L.fbun_tok = torch.empty(L.max_windows + 1, dtype=torch.int32, device=coors.device)
        L.num_fbundles = torch.empty(1, dtype=torch.int32, device=coors.device)
        j.fbun_tok, j.num_fbundles = L.fbun_tok.data_ptr(), L.num_fbundles.data_ptr()
        L.fitems = torch.empty((2 * (L.max_windows + 1), 4), dtype=torch.int32, device=coors.device)
        L.num_fitems = torch.empty(1, dtype=torch.int32, device=coors.device)
        j.fitems, j.num_fitems = L.fitems.data_ptr(), L.num_fitems.data_ptr()
        ns[k] = L.n
    wsb = lib.geomae_window_build_batch_workspace_bytes(ns, len(jobs), batch_size, ctypes.byref(wcfg))
    if wsb < 0:
        raise RuntimeError("window_build_batch: bad configuration (1..4 jobs)")
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=keep[0].device)
    check(lib.geomae_window_build_batch(arr, len(jobs), batch_size, ctypes.byref(wcfg), _ptr(ws), wsb, _stream()),
          "geomae_window_build_batch")
    return layouts


def window_build(coors, batch_size, wcfg, shift_index):
    coors = coors.contiguous()
    _check_input(coors, "coors", torch.int32)
    n = coors.shape[0]
    dev = coors.device
    lib = _lib.load()
    L = _new_window_layout(n, batch_size, wcfg, dev)
    wsb = lib.geomae_window_build_workspace_bytes(n, batch_size, ctypes.byref(wcfg))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    check(lib.geomae_window_build(_ptr(coors), n, batch_size, ctypes.byref(wcfg), shift_index, _ptr(L.win_start),
                                  _ptr(L.win_tokens), _ptr(L.tok_win), _ptr(L.tok_pos), _ptr(L.num_windows),
                                  _ptr(L.bun_start), _ptr(L.num_bundles), _ptr(ws), wsb, _stream()),
          "geomae_window_build")
    return L


class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, layout, num_heads):
        qkv = qkv.contiguous()
        _check_input(qkv, "qkv", torch.bfloat16)
        n, c3 = qkv.shape
        C = c3 // 3
        out = torch.empty((n, C), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((n, num_heads), dtype=torch.float32, device=qkv.device)
        with _timed("win_attn_fwd_kernel"):
            check(_lib.load().geomae_window_attention_forward(
                _ptr(qkv), n, num_heads, C // num_heads, _ptr(layout.win_start), _ptr(layout.win_tokens),
                _ptr(layout.tok_win), _ptr(layout.bun_start), _ptr(layout.num_bundles), layout.max_windows,
                layout.max_tokens, _ptr(out), _ptr(lse), _ptr(layout.bun_tok), _ptr(layout.pos_info), _stream()),
                "geomae_window_attention_forward")
        ctx.layout, ctx.num_heads = layout, num_heads
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        dout = dout.contiguous().to(torch.bfloat16)
        n, c3 = qkv.shape
        dqkv = torch.empty_like(qkv)
        L = ctx.layout
        with _timed("win_attn_bwd_kernel"):
            check(_lib.load().geomae_window_attention_backward(
                _ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), n, ctx.num_heads, c3 // 3 // ctx.num_heads,
                _ptr(L.win_start), _ptr(L.win_tokens), _ptr(L.tok_win), _ptr(L.bun_start), _ptr(L.num_bundles),
                L.max_windows, L.max_tokens, _ptr(dqkv), _ptr(L.bun_tok), _ptr(L.pos_info), _stream()),
                "geomae_window_attention_backward")
        return dqkv, None, None


def window_attention(qkv_bf16, layout, num_heads):
    """softmax(q k^T / sqrt(d)) v inside every window; qkv [n, 3C] bf16 -> [n, C] bf16."""
    return _WindowAttention.apply(qkv_bf16, layout, num_heads)


# ------------------------------------------------------------------------------------ operator-level window API
# mmdet3d/ops/__init__.py:22-26 exports these five names; other reference modules (SSTInputLayer, SST backbones) import
# them.  Same signatures and results as ops/sst/sst_ops.py; the index work is the pillar-segment counting sort and two
# row-copy kernels (csrc/winops.hip) instead of torch.sort / unique / bincount / cumsum chains.
def _window_rank(win_inds):
    """-> (continuous ids, rank inside the window), both int64 [N], on win_inds' device."""
    if win_inds.dim() != 1:
        raise RuntimeError("window indices must be a 1-d tensor")
    if not win_inds.is_cuda:
        raise RuntimeError("win_inds must be a CUDA tensor (geomae_amd has no CPU path)")
    n = win_inds.shape[0]
    dev = win_inds.device
    conti = torch.empty(n, dtype=torch.int64, device=dev)
    inner = torch.empty(n, dtype=torch.int64, device=dev)
    if n == 0:
        return conti, inner
    top = int(win_inds.max().item())                     # (the reference syncs for the same number, sst_ops.py:380)
    if int(win_inds.min().item()) < 0 or top >= 2 ** 31 - 1:
        raise RuntimeError("window indices must be in [0, 2^31 - 1)")
    coors = torch.zeros((n, 3), dtype=torch.int32, device=dev)
    coors[:, 2] = win_inds
    seg = pillar_segment_nd(coors, 1, (1, 1, top + 1))
    check(_lib.load().geomae_window_rank(_ptr(seg.order), _ptr(seg.inv), _ptr(seg.seg_start), n, _ptr(conti), _ptr(inner),
                                         _stream()), "geomae_window_rank")
    return conti, inner


@torch.no_grad()
def make_continuous_inds(inds):
    """sst_ops.py:371-388: relabel ids to 0..W-1 in ascending order of the original ids."""
    return _window_rank(inds)[0].to(inds.dtype)


@torch.no_grad()
def get_inner_win_inds(win_inds):
    """sst_ops.py:271-319: for the M tokens that share a window, a permutation of 0..M-1 (the reference's order follows
    an unstable sort, "might output different results" :280; any ranking is valid)."""
    return _window_rank(win_inds)[1].to(win_inds.dtype)


@torch.no_grad()
def get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug=True):
    """sst_ops.py:57-96: {level: (flat2window_inds [n_l], (positions of the level's tokens,))}."""
    out = {}
    for dl in drop_info:
        dl_mask = voxel_drop_lvl == dl
        if not dl_mask.any():
            continue
        conti, inner = _window_rank(batch_win_inds[dl_mask].contiguous())
        max_tokens = drop_info[dl]["max_tokens"]
        flat2window_inds = (conti * max_tokens + inner).to(batch_win_inds.dtype)
        out[dl] = (flat2window_inds, torch.where(dl_mask))
        if debug:
            assert int(inner.max()) < max_tokens, f"Max inner inds({int(inner.max())}) larger(equal) than {max_tokens}"
    return out


class _RowsCopy(torch.autograd.Function):
    """dst = zeros(rows_out); scatter: dst[idx[i]] = src[i]   |   gather: dst[i] = src[idx[i]]  (idx without repeats)."""

    @staticmethod
    def forward(ctx, src, idx, rows_out, scatter):
        src = src.contiguous()
        idx = idx.contiguous().long()
        if not src.is_cuda:
            raise RuntimeError("feat must be a CUDA tensor (geomae_amd has no CPU path)")
        n, row = idx.shape[0], src[0].numel() if src.shape[0] else int(torch.tensor(src.shape[1:]).prod())
        shape = (rows_out,) + tuple(src.shape[1:])
        dst = torch.zeros(shape, dtype=src.dtype, device=src.device) if scatter else \
            torch.empty(shape, dtype=src.dtype, device=src.device)
        fn = _lib.load().geomae_rows_scatter if scatter else _lib.load().geomae_rows_gather
        check(fn(_ptr(src), _ptr(idx), n, row * src.element_size(), _ptr(dst), _stream()), "geomae_rows_copy")
        ctx.save_for_backward(idx)
        ctx.rows_in, ctx.scatter = src.shape[0], scatter
        return dst

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _RowsCopy.apply(g, idx, ctx.rows_in, not ctx.scatter), None, None, None


def flat2window(feat, voxel_drop_lvl, flat2win_inds_dict, drop_info):
    """sst_ops.py:98-135: {level: zero-padded [num_windows, max_tokens, C]} (differentiable in feat)."""
    out = {}
    for dl in drop_info:
        dl_mask = voxel_drop_lvl == dl
        if not dl_mask.any():
            continue
        this_inds, where = flat2win_inds_dict[dl]
        max_tokens = drop_info[dl]["max_tokens"]
        num_windows = int((this_inds // max_tokens).max().item()) + 1
        rows = _RowsCopy.apply(feat, where[0], where[0].shape[0], False)                 # feat[dl_mask]
        feat_3d = _RowsCopy.apply(rows, this_inds, num_windows * max_tokens, True)
        out[dl] = feat_3d.reshape((num_windows, max_tokens) + tuple(feat.shape[1:]))
    return out


def window2flat(feat_3d_dict, inds_dict):
    """sst_ops.py:225-251: the inverse of flat2window -> [N, C]."""
    num_all = sum(inds_dict[dl][0].shape[0] for dl in inds_dict)
    first = feat_3d_dict[next(iter(feat_3d_dict))]
    out = None
    covered = 0
    for dl in feat_3d_dict:
        feat = feat_3d_dict[dl]
        inds, flat_pos = inds_dict[dl]
        rows = _RowsCopy.apply(feat.reshape(-1, feat.shape[-1]), inds, inds.shape[0], False)
        part = _RowsCopy.apply(rows, flat_pos[0], num_all, True)
        out = part if out is None else out + part
        covered += inds.shape[0]
    assert covered == num_all, "window2flat: some voxels belong to no level"          # the reference's check_feat
    return out if out is not None else first.new_zeros((num_all, first.shape[-1]))


# ------------------------------------------------------------------------------------ N1 fine-tune pieces
def window_drop(coors, batch_size, wcfg, shift_index, drop_info):
    """SSTInputLayer.drop_single_shift (sst_input_layer.py:213-238): -> keep [n] bool, drop_level [n] int32.
    drop_info: {level: dict(max_tokens=.., drop_range=(lower, upper))}."""
    coors = coors.contiguous()
    _check_input(coors, "coors", torch.int32)
    n, dev, lib = coors.shape[0], coors.device, _lib.load()
    keep = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    level = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    lv = sorted(drop_info)
    arr = lambda vals: (ctypes.c_int32 * len(vals))(*[int(min(v, 2 ** 31 - 1)) for v in vals])
    wsb = lib.geomae_window_drop_workspace_bytes(n, batch_size, ctypes.byref(wcfg))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    check(lib.geomae_window_drop(_ptr(coors), n, batch_size, ctypes.byref(wcfg), shift_index, len(lv),
                                 arr([drop_info[k]["max_tokens"] for k in lv]), arr([drop_info[k]["drop_range"][0] for k in lv]),
                                 arr([drop_info[k]["drop_range"][1] for k in lv]), _ptr(keep), _ptr(level), _ptr(ws), wsb,
                                 _stream()), "geomae_window_drop")
    return keep[:n].bool(), level[:n]


class _RecoverBev(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, coors, batch_size, ny, nx):
        feat = feat.contiguous().float()
        n, C = feat.shape
        canvas = torch.empty((batch_size, ny, nx, C), dtype=torch.float32, device=feat.device)
        check(_lib.load().geomae_recover_bev_forward(_ptr(feat), _ptr(coors), n, C, batch_size, ny, nx, _ptr(canvas),
                                                     _stream()), "geomae_recover_bev_forward")
        ctx.coors, ctx.shape = coors, (n, C, batch_size, ny, nx)
        return canvas.permute(0, 3, 1, 2)                     # [B, C, ny, nx] view in channels_last memory format

    @staticmethod
    def backward(ctx, g):
        n, C, B, ny, nx = ctx.shape
        g = g.permute(0, 2, 3, 1).contiguous().float()        # no copy when the conv backward is channels_last too
        out = torch.empty((n, C), dtype=torch.float32, device=g.device)
        check(_lib.load().geomae_recover_bev_backward(_ptr(g), _ptr(ctx.coors), n, C, B, ny, nx, _ptr(out), _stream()),
              "geomae_recover_bev_backward")
        return out, None, None, None, None


def recover_bev(feat, coors, batch_size, ny, nx):
    """SSTSecondPretrainedv1.recover_bev (sst_second_pretrained_v1.py:243-280): [n,C] tokens -> dense [B,C,ny,nx]."""
    coors = coors.contiguous()
    _check_input(coors, "coors", torch.int32)
    return _RecoverBev.apply(feat, coors, batch_size, ny, nx)


# ------------------------------------------------------------------------------------ fused SST layer
def pack_weights(desc, num_desc, max_elems, packed, aux=None):
    check(_lib.load().geomae_pack_weights(ctypes.c_void_p(0), _ptr(desc), num_desc, max_elems, _ptr(packed),
                                          _ptr(aux), _stream()), "geomae_pack_weights")


def heads_loss(cen, den, n_keep, n_mask, head_w, head_bias, tgt, weights, d_out=None, losses=None, split=False):
    """-> losses [6] f32, d_cen, d_den [n,128] f32 (gradient of sum(losses)), saved = (dlogits, cm_b, dm_b).
    d_out: optional ZEROED [n,128] f32 buffers for d_cen / d_den (the kernel writes only the masked rows).
    losses: optional ZEROED [6] f32 buffer (then no memset is enqueued in front of the kernel).
    split: the two-workgroups-per-tile form (geomae_heads_loss_split_accumulate): d(centroid decoder output) comes back as
    TWO summands -- d_cen is then the pair (d_cen, d_cen2), to be handed to sst_stack_backward as dz and dz_add -- and
    d_out, when given, holds three buffers (d_cen, d_cen2, d_den)."""
    dev = cen.device
    n = cen.shape[0]
    lib = _lib.load()
    if losses is None:
        losses = torch.zeros(6, dtype=torch.float32, device=dev) if split else torch.empty(6, dtype=torch.float32, device=dev)
        fn = lib.geomae_heads_loss
    else:
        fn = lib.geomae_heads_loss_accumulate
    want = 3 if split else 2
    if d_out is None:
        outs = [torch.zeros_like(cen) for _ in range(want - 1)] + [torch.zeros_like(den)]
    else:
        outs = list(d_out)
        if len(outs) != want or any(t.shape != cen.shape or not t.is_contiguous() for t in outs):
            raise RuntimeError(f"heads_loss: d_out must be {want} contiguous buffers of the decoder outputs' shape")
    dl = torch.empty((n_mask, 896), dtype=torch.bfloat16, device=dev)
    cm_b = torch.empty((n_mask, 128), dtype=torch.bfloat16, device=dev)
    dm_b = torch.empty((n_mask, 128), dtype=torch.bfloat16, device=dev)
    head = (_ptr(cen), _ptr(den), n_keep, n_mask, _ptr(head_w), _ptr(head_bias), _ptr(tgt["centroid_low"]),
            _ptr(tgt["mask_low_u8"]), _ptr(tgt["centroid_med"]), _ptr(tgt["mask_med_u8"]), _ptr(tgt["centroid_top"]),
            _ptr(tgt["normal"]), _ptr(tgt["occ_counts"]), f3(weights), _ptr(losses))
    tail = (_ptr(dl), _ptr(cm_b), _ptr(dm_b), _stream())
    if split:
        check(lib.geomae_heads_loss_split_accumulate(*head, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), *tail),
              "geomae_heads_loss_split_accumulate")
        return losses, (outs[0], outs[1]), outs[2], (dl, cm_b, dm_b)
    check(fn(*head, _ptr(outs[0]), _ptr(outs[1]), *tail), "geomae_heads_loss")
    return losses, outs[0], outs[1], (dl, cm_b, dm_b)


def heads_loss_by_decoder(cen, den, n_keep, n_mask, head_w, head_bias, tgt, weights, side_stream=None):
    """The heads as TWO launches with disjoint outputs (geomae_heads_loss_centroid_accumulate: the five heads that read
    the centroid decoder; geomae_heads_loss_density_accumulate: the density decoder's normal head) -- what the step
    engine runs, each on its decoder's stream.  side_stream: where to launch the density part (default: current).
    -> losses [6], (d_cen, d_cen2), d_den, (dlogits, cm_b, dm_b): as heads_loss(split=True)."""
    dev = cen.device
    lib = _lib.load()
    losses = torch.zeros(6, dtype=torch.float32, device=dev)
    d_cen, d_cen2, d_den = torch.zeros_like(cen), torch.zeros_like(cen), torch.zeros_like(den)
    dl = torch.empty((n_mask, 896), dtype=torch.bfloat16, device=dev)
    cm_b = torch.empty((n_mask, 128), dtype=torch.bfloat16, device=dev)
    dm_b = torch.empty((n_mask, 128), dtype=torch.bfloat16, device=dev)
    check(lib.geomae_heads_loss_centroid_accumulate(
        _ptr(cen), n_keep, n_mask, _ptr(head_w), _ptr(head_bias), _ptr(tgt["centroid_low"]), _ptr(tgt["mask_low_u8"]),
        _ptr(tgt["centroid_med"]), _ptr(tgt["mask_med_u8"]), _ptr(tgt["centroid_top"]), _ptr(tgt["occ_counts"]), f3(weights),
        _ptr(losses), _ptr(d_cen), _ptr(d_cen2), _ptr(dl), _ptr(cm_b), _stream()), "geomae_heads_loss_centroid_accumulate")
    cur = torch.cuda.current_stream(dev)
    side = side_stream if side_stream is not None else cur
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        check(lib.geomae_heads_loss_density_accumulate(
            _ptr(den), n_keep, n_mask, _ptr(head_w), _ptr(head_bias), _ptr(tgt["normal"]), f3(weights), _ptr(losses),
            _ptr(d_den), _ptr(dl), _ptr(dm_b), _stream()), "geomae_heads_loss_density_accumulate")
    cur.wait_stream(side)
    return losses, (d_cen, d_cen2), d_den, (dl, cm_b, dm_b)


def heads_weight_grad(n_mask, dl, cm_b, dm_b, grads):
    check(_lib.load().geomae_heads_weight_grad(n_mask, _ptr(dl), _ptr(cm_b), _ptr(dm_b), ctypes.byref(grads),
                                               _stream()), "geomae_heads_weight_grad")


# ------------------------------------------------------------------------------------ fused VFE
VFE_MOMENTS = True       # False: the sweep forms of the layer-0 statistics / weight gradient (A/B, tests)
VFE_PILLAR_STATS = True  # False: the layer-1 BatchNorm-backward sums always by the sweep over the points (A/B, tests)


def vfe_prepare_points(points, seg, voxel_size, center_offset, zeros=None):
    """The weight-independent front of the fused VFE: pillar means and the decorated point features in pillar order
    (-> (mean [cap,3], feat [N,16], pid [N])).  It needs only the points and their segments, not the pillar count on
    the host, so a training loop can run it for the NEXT batch together with that batch's voxelization."""
    lib = _lib.load()
    dev, N = points.device, points.shape[0]
    mean = segment_mean_xyz(points, seg, zeros)
    feat = torch.empty((max(N, 1), 16), dtype=torch.float32, device=dev)
    pid = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    # ... and the moments of the decorated features (fp64 [16 + 121]): layer 0 is linear, so its BatchNorm statistics and
    # the BatchNorm term of its weight gradient follow from them without a sweep over the points (csrc/vfe.hip)
    moments = torch.zeros(144, dtype=torch.float64, device=dev)
    ws = torch.empty(lib.geomae_vfe_moments_workspace_bytes(), dtype=torch.uint8, device=dev)
    check(lib.geomae_vfe_prepare_moments(_ptr(points), points.shape[1], N, _ptr(seg.order), _ptr(seg.inv), _ptr(mean),
                                         _ptr(seg.voxel_coors), f3(voxel_size), f3(center_offset), _ptr(feat), _ptr(pid),
                                         _ptr(ws), _ptr(moments), _stream()), "geomae_vfe_prepare_moments")
    return mean, feat, pid, moments


class VfePlan:
    """Per-batch state of the fused VFE sweeps: pillar mean, sorted point features, the argument struct."""

    def __init__(self, points, seg, w0, w1, voxel_size, center_offset, zeros=None, prepared=None, layer1_bf16=False):
        from ._lib import GeomaeVfeArgs
        dev = points.device
        self.points, self.seg, self.N, self.V = points, seg, points.shape[0], seg.V
        if prepared is None:
            prepared = vfe_prepare_points(points, seg, voxel_size, center_offset, zeros)
        self.mean, self.feat, self.pid = prepared[:3]
        self.moments = prepared[3] if len(prepared) > 3 and VFE_MOMENTS else None
        self.dw0_acc = None
        # [layer][scale, shift, mean, invstd]
        self.bn = zeros.take((2, 4, 128), torch.float32) if zeros is not None else \
            torch.zeros((2, 4, 128), dtype=torch.float32, device=dev)
        a = GeomaeVfeArgs()
        a.feat_sorted, a.pid_sorted, a.seg_start = self.feat.data_ptr(), self.pid.data_ptr(), seg.seg_start.data_ptr()
        a.num_points, a.max_pillars = self.N, max(self.V, 1)
        a.w0, a.w1 = w0.data_ptr(), w1.data_ptr()
        a.scale0, a.shift0 = self.bn[0, 0].data_ptr(), self.bn[0, 1].data_ptr()
        a.scale1, a.shift1 = self.bn[1, 0].data_ptr(), self.bn[1, 1].data_ptr()
        if self.moments is not None:
            a.moments = self.moments.data_ptr()
        a.layer1_bf16 = int(bool(layer1_bf16))     # plain bf16 layer-1 products (the bf16 compute mode) | bf16 x 3, fp32 grade
        self.args = a
        self._keep = (w0, w1)

    def bn_state(self):
        from ._lib import GeomaeBnState
        b = GeomaeBnState()
        for layer in (0, 1):
            for k, name in enumerate(("scale", "shift", "mean", "invstd")):
                setattr(b, f"{name}{layer}", self.bn[layer, k].data_ptr())
        return b


# Process group of the fused VFE's BatchNorm collectives when the caller passes none (None = the default group).  The
# trainer installs one of its own at world > 1: on the default group's communicator the [2C]-word all-reduces of the VFE
# backward would queue behind the encoder segment's gradient all-reduce that was started just before it.
BN_GROUP = None
GRAD_GROUP = None      # the gradient segments' communicator (train.Trainer creates both at world > 1)


def _bn_finalize(plan, layer, sums, norm, world, group):
    """Training-mode statistics -> folded scale/shift (+ running stats), with naiveSyncBN1d's equal-weight
    cross-rank average of (mean, mean of squares) when world > 1 (mmdet3d/ops/norm.py:64-76)."""
    from torch import distributed as dist
    group = BN_GROUP if group is None else group
    lib = _lib.load()
    C = 64 if layer == 0 else 128
    bn = plan.bn[layer]
    common = (C, _ptr(norm.weight), _ptr(norm.bias), float(norm.eps), float(norm.momentum))
    if world == 1:
        check(lib.geomae_bn_finalize(_ptr(sums), float(plan.N), None, *common, 1, _ptr(norm.running_mean),
                                     _ptr(norm.running_var), _ptr(bn[0]), _ptr(bn[1]), _ptr(bn[3]), _ptr(bn[2]),
                                     _ptr(norm.num_batches_tracked), _stream()), "geomae_bn_finalize")
    else:
        mom = torch.empty(2 * C, dtype=torch.float32, device=sums.device)
        # (divisor N * world: the SUM all-reduce then delivers the equal-weight average of the ranks' moments directly)
        check(lib.geomae_bn_finalize(_ptr(sums), float(plan.N) * world, None, C, None, None, 0.0, 0.0, 0, None, None, None, None,
                                     None, _ptr(mom), None, _stream()), "geomae_bn_finalize")
        dist.all_reduce(mom, group=group)
        # (no batch counter here: the reference's cross-rank branch never touches num_batches_tracked, ops/norm.py:58-86)
        check(lib.geomae_bn_finalize(None, float(plan.N), _ptr(mom), *common, 0, _ptr(norm.running_mean),
                                     _ptr(norm.running_var), _ptr(bn[0]), _ptr(bn[1]), _ptr(bn[3]), _ptr(bn[2]),
                                     None, _stream()), "geomae_bn_finalize")


def vfe_forward_zero_specs(cap, V, prepared=False):
    """(shape, dtype) of the buffers VfePlan + vfe_forward carve from a ZeroArena, in carve order."""
    V1 = max(int(V), 1)
    return [((2, 4, 128), torch.float32), ((128,), torch.float64), ((V1, 64), torch.float32),
                    ((256,), torch.float64), ((V1, 128), torch.float32), ((V1,), torch.uint8)]


def vfe_backward_zero_specs(V):
    return [((256,), torch.float64), ((max(int(V), 1), 64), torch.float32), ((128,), torch.float64)]


def vfe_forward(plan, norm0, norm1, world=1, group=None, zeros=None):
    lib = _lib.load()
    dev = plan.points.device
    a = ctypes.byref(plan.args)
    sums0 = _zeros_or_empty(zeros, (128,), torch.float64, dev)
    check(lib.geomae_vfe_stats0(a, _ptr(sums0), _stream()), "geomae_vfe_stats0")
    _bn_finalize(plan, 0, sums0, norm0, world, group)
    m0 = _zeros_or_empty(zeros, (max(plan.V, 1), 64), torch.float32, dev)
    sums1 = _zeros_or_empty(zeros, (256,), torch.float64, dev)
    check(lib.geomae_vfe_layer0(a, _ptr(m0), _ptr(sums1), _stream()), "geomae_vfe_layer0")
    _bn_finalize(plan, 1, sums1, norm1, world, group)
    vf = _zeros_or_empty(zeros, (max(plan.V, 1), 128), torch.float32, dev)
    # a byte per pillar that the layer-1 sweep sets where two points may share the pillar's maximum: for the others the
    # backward's BatchNorm sums come from the pillar rows alone (GeomaeVfeArgs.pillar_ties)
    plan.pillar_ties = _zeros_or_empty(zeros, (max(plan.V, 1),), torch.uint8, dev)
    plan.args.pillar_ties = plan.pillar_ties.data_ptr() if VFE_PILLAR_STATS else None
    check(lib.geomae_vfe_layer1(a, _ptr(m0), _ptr(vf), _stream()), "geomae_vfe_layer1")
    return vf[:plan.V], m0


def vfe_backward(plan, m0, vf, dvf, params, world=1, group=None, zeros=None, side=None):
    """params: dict w0, w1, g0, b0, g1, b1 -> nn.Parameters whose .grad is accumulated into.
    zeros: optional ZeroArena sized by vfe_backward_zero_specs (call under prezeroed()).
    side: optional stream for the layer-1 weight-gradient contraction (read only by the optimizer): it then runs beside
    the layer-0 backward kernels; the caller joins `side` before the optimizer."""
    from torch import distributed as dist
    lib = _lib.load()
    dev = dvf.device
    a, bn = ctypes.byref(plan.args), plan.bn_state()
    N, V = plan.N, plan.V
    for p in params.values():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    dvf = dvf.contiguous().float()
    bs1 = _zeros_or_empty(zeros, (256,), torch.float64, dev)
    check(lib.geomae_vfe_backward_stats(a, ctypes.byref(bn), _ptr(m0), _ptr(vf), _ptr(dvf), _ptr(bs1), _stream()),
          "geomae_vfe_backward_stats")
    # d beta / d gamma = the LOCAL sums: one process lets the next kernel add them; with naiveSyncBN1d they are added
    # here, before the sums are all-reduced for the input gradient
    fold = world == 1
    group = BN_GROUP if group is None else group
    if not fold:
        params["b1"].grad.add_(bs1[:128])
        params["g1"].grad.add_(bs1[128:])
        dist.all_reduce(bs1, group=group)
    n_eff = float(world * N)
    dy1_b = torch.empty(((N + 15) // 16 * 16, 128), dtype=torch.bfloat16, device=dev)     # tile-blocked scratch operands
    g_b = torch.empty(((N + 15) // 16 * 16, 128), dtype=torch.bfloat16, device=dev)
    dh0 = torch.empty((N, 64), dtype=torch.float32, device=dev)
    dm0 = _zeros_or_empty(zeros, (max(V, 1), 64), torch.float32, dev)
    bs0 = _zeros_or_empty(zeros, (128,), torch.float64, dev)
    if plan.moments is not None:          # layer-0 weight gradient from one contraction + the moments: no sweep of its own
        plan.dw0_acc = torch.zeros(64 * 16, dtype=torch.float32, device=dev)
        plan.args.dw0_acc = plan.dw0_acc.data_ptr()
    check(lib.geomae_vfe_backward_layer1(a, ctypes.byref(bn), _ptr(m0), _ptr(vf), _ptr(dvf), _ptr(bs1), n_eff,
                                         _ptr(dy1_b), _ptr(g_b), None, _ptr(dh0), _ptr(dm0), _ptr(bs0),
                                         _ptr(params["b1"].grad) if fold else None,
                                         _ptr(params["g1"].grad) if fold else None, _stream()), "geomae_vfe_backward_layer1")
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        dy1_b.record_stream(side)
        g_b.record_stream(side)
        with torch.cuda.stream(side):
            check(lib.geomae_vfe_weight_grad1(_ptr(dy1_b), _ptr(g_b), N, _ptr(params["w1"].grad), _stream()),
                  "geomae_vfe_weight_grad1")
    if not fold:
        params["b0"].grad.add_(bs0[:64])
        params["g0"].grad.add_(bs0[64:])
        dist.all_reduce(bs0, group=group)
    check(lib.geomae_vfe_backward_layer0(a, ctypes.byref(bn), _ptr(dh0), _ptr(bs0), n_eff, N, _ptr(dy1_b), _ptr(g_b),
                                         _ptr(params["w0"].grad), None if side is not None else _ptr(params["w1"].grad),
                                         _ptr(params["b0"].grad) if fold else None,
                                         _ptr(params["g0"].grad) if fold else None, _stream()),
          "geomae_vfe_backward_layer0")


# ------------------------------------------------------------------------------------ SST layer stacks
PROFILER = None        # bench.py: handle from geomae_profiler_create, passed to every stack call
KERNEL_IDS = dict(sst_qkv_fwd_kernel=1, win_attn_fwd_kernel=2, sst_ffn_fwd_kernel=3, sst_ffn_bwd_kernel=4,
                  win_attn_bwd_kernel=5, sst_qkv_bwd_kernel=6, dw_kernel=7, sst_ffn_bwd_dw_kernel=8,
                  sst_ffn_fwd_pair_kernel=9, sst_layer_fwd_kernel=10, sst_layer_bwd_kernel=11)


def _stack_layouts(layouts):
    from ._lib import GeomaeSstStackLayout
    arr = (GeomaeSstStackLayout * 2)()
    for i in range(2):
        L = layouts[i % len(layouts)]
        a = arr[i]
        a.win_start, a.win_tokens, a.tok_win = L.win_start.data_ptr(), L.win_tokens.data_ptr(), L.tok_win.data_ptr()
        a.tok_pos, a.bun_start, a.num_bundles = L.tok_pos.data_ptr(), L.bun_start.data_ptr(), L.num_bundles.data_ptr()
        a.max_bundles = L.max_windows
        if L.bun_tok is not None and L.pos_info is not None:
            a.bun_tok, a.pos_info = L.bun_tok.data_ptr(), L.pos_info.data_ptr()
        if getattr(L, "fbun_tok", None) is not None and L.num_fbundles is not None:
            a.fbun_tok, a.num_fbundles = L.fbun_tok.data_ptr(), L.num_fbundles.data_ptr()
        if getattr(L, "fitems", None) is not None and L.num_fitems is not None:
            a.fitems, a.num_fitems = L.fitems.data_ptr(), L.num_fitems.data_ptr()
    return arr


def _stream_of(stream):
    return _stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)


def sst_stack_forward(x, weights, layouts, pos_table, num_heads, stream=None, out=None, tail=None, rows=None, big_layouts=None):
    """weights: ctypes array of GeomaeSstLayerWeights (one per layer).  -> z [n,128] f32, saved blob (uint8).
    Buffers are allocated on the CURRENT stream; the kernels are enqueued on `stream` (default: current).
    out: optional preallocated contiguous [n,128] f32 destination of z.
    tail = (fill_row [1,128] or [128] f32, count): the stack's input is x followed by `count` copies of fill_row (the
    decoders' mask tokens), without materialising them; n = x.shape[0] + count.
    rows: optional int32 [m] row indices into x: the stack's input is x[rows] (gathered by its input conversion)."""
    lib = _lib.load()
    _check_input(x, "x", torch.float32)
    n_in, nl = x.shape[0], len(weights)
    if rows is not None:
        _check_input(rows, "rows", torch.int32)
        n_in = rows.numel()
    fill, n = None, n_in
    if tail is not None:
        fill, extra = tail
        _check_input(fill, "fill_row", torch.float32)
        if fill.numel() != x.shape[1]:
            raise RuntimeError("sst_stack_forward: fill_row must have one row of x's width")
        n = n_in + int(extra)
    if big_layouts is not None:       # the caller's promise for this call: which layouts may hold a bundle of more than four tiles
        lib.geomae_sst_set_big_bundle_layouts(int(big_layouts))
    sb = lib.geomae_sst_stack_saved_bytes(n, nl, num_heads)
    saved = torch.empty(max(sb, 1), dtype=torch.uint8, device=x.device)
    if out is None:
        z = torch.empty((n, x.shape[1]), dtype=torch.float32, device=x.device)
    else:
        _check_input(out, "out", torch.float32)
        if out.shape != (n, x.shape[1]):
            raise RuntimeError("sst_stack_forward: out must be [n, width of x]")
        z = out
    check(lib.geomae_sst_stack_forward(_ptr(x), n, weights, nl, _stack_layouts(layouts), _ptr(pos_table), num_heads,
                                       layouts[0].max_tokens, _ptr(saved), sb, _ptr(z), n_in, _ptr(fill), _ptr(rows),
                                       ctypes.c_void_p(PROFILER) if PROFILER else None, _stream_of(stream)),
          "geomae_sst_stack_forward")
    return z, saved


def flush_weight_grad(stream=None):
    """Launch the weight-gradient contraction a stack backward left recorded (defer_last=True) on `stream`."""
    check(_lib.load().geomae_flush_weight_grad(_stream_of(stream)), "geomae_flush_weight_grad")


def last_stack_forms():
    """(forward, backward) form of the last sst_stack_forward / sst_stack_backward of this thread: 0 three-launch, 1 one-launch,
    2 looping (include/geomae_hip.h GEOMAE_STACK_FORM_*), -1 none yet."""
    out = (ctypes.c_int32 * 2)()
    check(_lib.load().geomae_sst_last_stack_forms(out), "geomae_sst_last_stack_forms")
    return int(out[0]), int(out[1])


def fused_dropped_bundles(reset=True):
    """Bundles of more than four tiles that no one-launch kernel ran since the last reset (a broken big-bundle promise); syncs."""
    out = ctypes.c_int64(0)
    check(_lib.load().geomae_sst_fused_dropped_bundles(ctypes.byref(out), int(bool(reset))), "geomae_sst_fused_dropped_bundles")
    return int(out.value)


def sst_stack_backward(dz, n, weights, grads, layouts, pos_table, num_heads, saved, stream=None, defer_last=False,
                       scatter=None, dz_add=None, tail_sum=None, defer_all=False, big_layouts=None):
    """defer_last: leave the first layer's weight-gradient contraction recorded (-> also returns the scratch buffer,
    which must stay alive until flush_weight_grad's kernel ran).
    scatter = (rows int32 [n], dst [m,128] f32): the input gradient of token t is written to dst[rows[t]] (the
    transpose of sst_stack_forward's `rows`; dst's other rows are left as they are) and dst is returned as dx.
    dz_add: optional second summand of the output gradient (>= n rows; the stack reads dz + dz_add).
    tail_sum = (acc [128] or [1,128] f32, from_row): the column sums of dx[from_row:] are ADDED into acc (the gradient of
    sst_stack_forward's fill row).
    defer_all: EVERY layer's contraction is left recorded for flush_weight_grad (the step engine's mode; one slab set per layer;
    -> returns (dx, scratch)).  big_layouts: the caller's promise for this call (geomae_sst_set_big_bundle_layouts): bit s =
    layout s may hold a bundle of more than four tiles (None: both may)."""
    lib = _lib.load()
    _check_input(dz, "dz", torch.float32)
    nl = len(weights)
    wb = lib.geomae_sst_stack_scratch_bytes_layers(n, nl) if defer_all else lib.geomae_sst_stack_scratch_bytes(n)
    if big_layouts is not None:
        lib.geomae_sst_set_big_bundle_layouts(int(big_layouts))
    scratch = torch.empty(max(wb, 1), dtype=torch.uint8, device=dz.device)
    rows, n_out = None, 0
    if scatter is not None:
        rows, dx = scatter
        _check_input(rows, "rows", torch.int32)
        _check_input(dx, "dst", torch.float32)
        if rows.numel() != n or dx.dim() != 2 or dx.shape[1] != dz.shape[1]:
            raise RuntimeError("sst_stack_backward: scatter needs one row index per token and a [m, width] destination")
        n_out = dx.shape[0]
    else:
        dx = dz.new_empty((n,) + tuple(dz.shape[1:]))
    tsum, tfrom = None, 0
    if tail_sum is not None:
        tsum, tfrom = tail_sum
        _check_input(tsum, "tail_sum", torch.float32)
        if tsum.numel() != dz.shape[1] or not 0 <= int(tfrom) <= n:
            raise RuntimeError("sst_stack_backward: tail_sum needs one accumulator per channel and 0 <= from_row <= n")
    if dz_add is not None:
        _check_input(dz_add, "dz_add", torch.float32)
        if dz_add.shape[0] < n or dz_add.shape[1:] != dz.shape[1:]:
            raise RuntimeError("sst_stack_backward: dz_add must hold at least n rows of dz's width")
    check(lib.geomae_sst_stack_backward(_ptr(dz), _ptr(dz_add), n, weights, grads, nl, _stack_layouts(layouts), _ptr(pos_table),
                                        num_heads, layouts[0].max_tokens, _ptr(saved), _ptr(scratch), wb, _ptr(dx),
                                        _ptr(rows), n_out, _ptr(tsum), int(tfrom), 2 if defer_all else int(bool(defer_last)), ctypes.c_void_p(PROFILER) if PROFILER else None,
                                        _stream_of(stream)),
          "geomae_sst_stack_backward")
    # `scratch` must outlive the kernels: a caller that runs the stack on a stream of its own keeps it until the join
    return (dx, scratch) if (stream is not None or defer_last or defer_all) else dx
