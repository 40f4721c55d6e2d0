"""ffn backward on the SAME buffers repeatedly (warm TLB / MALL) vs. rotating through many buffers (cold)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from geomae_amd import _lib
lib = _lib.load(path='/root/repo/tools/libgeomae_timing.so')
import geomae_amd
from geomae_amd.configs import mae_sst_model
from geomae_amd.sst import PackedLayers
dev = torch.device('cuda:0')
cfg = mae_sst_model(); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
bb = model.backbone
packed = bb._packed if hasattr(bb, '_packed') else None
import inspect
for name in dir(bb):
    v = getattr(bb, name, None)
    if isinstance(v, PackedLayers): packed = v
assert packed is not None, [n for n in dir(bb) if 'pack' in n.lower()]
packed.refresh()
lib.geomae_debug_read_stamps.restype = ctypes.c_int
lib.geomae_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
SL, NB = 32, 512
def read():
    buf = np.zeros(NB * SL, dtype=np.uint64)
    lib.geomae_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
    return buf.reshape(NB, SL).astype(np.int64)
P = lambda t: ctypes.c_void_p(t.data_ptr())
w = packed.weight_array(0, 1); g = packed.grad_array(0, 1)
def mk(n):
    f = lambda c, dt=torch.float32: torch.randn(n, c, device=dev).to(dt)
    return dict(xh1=f(128), xh2=f(128), hp=f(256, torch.bfloat16), rstd=torch.rand(n, 2, device=dev) + 0.5, dz=f(128),
                dx=f(128), dattn=f(128, torch.bfloat16), du=f(128, torch.bfloat16), dv=f(128, torch.bfloat16),
                dhp=f(256, torch.bfloat16), y=f(128, torch.bfloat16), h=f(256, torch.bfloat16))
def run(b, n):
    rc = lib.geomae_sst_ffn_backward(P(b['xh1']), P(b['xh2']), P(b['hp']), P(b['rstd']), P(b['dz']), w, n, P(b['dx']), P(b['dattn']),
                                     P(b['du']), P(b['dv']), P(b['dhp']), P(b['y']), P(b['h']), g, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
names = {1: 'ld+LN2bwd+st', 2: 'w2T issue', 3: 'bar1', 4: 'lds+bar2', 5: 'mfma', 6: 'hp+gelu+st', 7: 'w1T issue', 8: 'bar1', 9: 'lds+bar2', 10: 'mfma',
         11: 'LN1bwd+st', 12: 'woT issue', 13: 'bar1', 14: 'lds+bar2', 15: 'mfma', 16: 'st dattn', 17: 'flush'}
def report(tag, n):
    st = read(); nb = min((n + 63) // 64, NB); s = st[:nb]; rel = s - s[:, :1]
    line = [f"{tag:22s} n={n:6d} total med {np.median(rel[:,17]):7.0f} cyc | "]
    last = np.zeros(nb)
    for k in sorted(names):
        line.append(f"{names[k]}={np.median(rel[:,k]-last):.0f}")
        last = rel[:, k]
    print(" ".join(line))
    ev = None
for n in (6700, 22000, 64 * 2048):
    b = mk(n)
    for i in range(5): run(b, n)
    read(); run(b, n); report("warm same buffers", n)
    pool = [mk(n) for _ in range(40 if n < 50000 else 4)]     # 40 x ~ (n * 4.5 KB) : >> 256 MB MALL for n = 22000
    for pb in pool: run(pb, n)
    read(); run(pool[0], n); report("cold (rotated pool)", n)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(20): run(b, n)
    e.record(); torch.cuda.synchronize(); print("   warm avg us", s.elapsed_time(e) * 1000 / 20)
    s.record()
    for i in range(len(pool)): run(pool[i], n)
    e.record(); torch.cuda.synchronize(); print("   cold avg us", s.elapsed_time(e) * 1000 / len(pool))
    del pool
