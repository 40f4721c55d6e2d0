import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import geomae_oracle as O
from geomae_amd import ops, synth
from test_gpu_parity import _ref_window_attention, LEVELS, RANGE
dev=torch.device('cuda:0')
frames = [synth.lidar_frame(21), synth.lidar_frame(22, beams=16, n_az=300)]
_, coors = O.voxelize_batch(frames, LEVELS["top"], RANGE)
vc = O.unique_rows(coors)[0]
wcfg = ops.make_window_config((12, 12), (6, 6), (400, 400))
L = ops.window_build(torch.as_tensor(vc, device=dev), 2, wcfg, 0)
win = O.window_partition(vc, (12, 12), [(0, 0), (6, 6)], LEVELS["top"], RANGE)[0][0][0]
n = vc.shape[0]
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(n, 384, generator=g) * 1.5).bfloat16()
want = _ref_window_attention(qkv.double(), win, 8).float().numpy()
# pollute LDS / registers with NaN patterns: reductions and sorts over NaN tensors
junk = torch.full((4096, 4096), float('nan'), device=dev)
for _ in range(3):
    junk.sum(1); torch.softmax(junk, 1); torch.sort(junk[:256], dim=1); (junk.bfloat16() @ junk.bfloat16()[:, :256])
torch.cuda.synchronize()
uniq, inv = np.unique(win, return_inverse=True)
sizes = np.bincount(inv); wsz = sizes[inv]
for rep in range(3):
    got = ops.window_attention(qkv.to(dev), L, 8).float().cpu().numpy()
    nanrow = np.isnan(got).any(1)
    print("rep", rep, "nan rows", nanrow.sum(), "bad rows", (np.abs(np.nan_to_num(got)-want).max(1)>0.05).sum())
    if nanrow.any():
        rows = np.nonzero(nanrow)[0]
        print(" window sizes of nan rows:", np.unique(wsz[rows], return_counts=True))
        cols = np.isnan(got[rows]).any(0)
        print(" nan heads:", np.unique(np.nonzero(cols)[0] // 16))
        wt = L.win_tokens.cpu().numpy(); ws = L.win_start.cpu().numpy()
        pos = np.empty(n, int); pos[wt[:n]] = np.arange(n)
        w0 = inv[rows[0]]
        print(" first nan row", rows[0], "window size", wsz[rows[0]], "positions in window of nan rows of that window:",
              sorted((pos[r] - ws[L.tok_win.cpu().numpy()[r]]) for r in rows if inv[r] == w0)[:40])
