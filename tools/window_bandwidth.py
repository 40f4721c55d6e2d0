"""HBM traffic of a WINDOW of a training step: which dispatches ran inside it (a plain rocprofv3 kernel trace: real
concurrency, real durations) x what each of them moves (separate FETCH_SIZE / WRITE_SIZE passes aggregated per kernel and
grid size, tools/pmc.sh with PMC_BY_GRID=1; counters serialise the kernels, so the bytes come from there and the time from
the plain trace) / the window's duration.  Windows: the decoders' forward, the decoders' backward (first ffn-backward launch
of a decoder-size grid .. last decoder-size attention-backward), the encoder backward, the whole step.
usage: window_bandwidth.py kernel_trace.csv fetch_summary.csv write_summary.csv [+step]"""
import csv, re, sys
trace, fsum, wsum = sys.argv[1:4]
sel = sys.argv[4] if len(sys.argv) > 4 else "+8"


def key(name, grid):
    return name[:48] + " g" + str(grid)


def load(path, col):
    d = {}
    for r in csv.DictReader(open(path)):
        d[r["kernel"]] = float(r[col])
    return d
F, W = load(fsum, "FETCH_SIZE"), load(wsum, "WRITE_SIZE")
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
steps, prev = [], None
for i in ends:
    if prev is not None and i - prev > 5:
        steps.append((prev + 1, i))
    prev = i
lo, hi = steps[int(sel[1:])]
step = rows[lo:hi + 1]


def nbytes(r):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    k = key(r["Kernel_Name"].replace(",", ";"), g)
    if k in F and k in W:
        return (2 * F[k] + W[k]) * 1024.0            # the guide's gfx950 correction (FETCH_SIZE counts half of a wide read stream)
    # (grid sizes differ between the runs by a few workgroups: the nearest grid of the same kernel)
    cands = [kk for kk in F if kk.split(" g")[0] == k.split(" g")[0] and kk in W]
    if not cands:
        return 0.0
    kk = min(cands, key=lambda c: abs(int(c.split(" g")[1]) - g))
    return (2 * F[kk] + W[kk]) * 1024.0


def wgs(r):
    return (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))


def short(r):
    return re.sub(r"\(.*", "", r["Kernel_Name"]).replace("geomae::", "").replace("void ", "")


def window(label, first, last):
    if first is None or last is None:
        print(f"{label}: not found")
        return
    t0, t1 = int(first["Start_Timestamp"]), int(last["End_Timestamp"])
    tot, per = 0.0, {}
    for r in rows:                                   # every dispatch of the run that overlaps the window (any queue)
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ov = min(e, t1) - max(s, t0)
        if ov <= 0 or e <= s:
            continue
        b = nbytes(r) * ov / (e - s)
        tot += b
        per[short(r)] = per.get(short(r), 0.0) + b
    us = (t1 - t0) / 1e3
    print(f"{label}: {us:.1f} us, {tot / 1e6:.1f} MB of HBM traffic (FETCH + WRITE counters) -> {tot / us / 1e6:.2f} TB/s = {tot / us / 1e6 / 8.0 * 100:.0f} % of 8 TB/s")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]:
        print(f"      {k:36s} {v / 1e6:8.1f} MB")


dec_size = max(wgs(r) for r in step if "sst_ffn_bwd_kernel" in r["Kernel_Name"])
ffn_f = [r for r in step if "sst_ffn_fwd_kernel" in r["Kernel_Name"]]
ffn_b = [r for r in step if "sst_ffn_bwd_kernel" in r["Kernel_Name"] and wgs(r) == dec_size]
att_b = [r for r in step if "win_attn_bwd_kernel" in r["Kernel_Name"]]
big_att = max(wgs(r) for r in att_b)
dec_att_b = [r for r in att_b if wgs(r) == big_att]
enc_att_b = [r for r in att_b if wgs(r) != big_att]
qkv_f = [r for r in step if "sst_qkv_fwd_kernel" in r["Kernel_Name"]]
window("decoders forward ", qkv_f[0] if qkv_f else None, ffn_f[-1] if ffn_f else None)
window("decoders backward", ffn_b[0] if ffn_b else None, dec_att_b[-1] if dec_att_b else None)
lay_b = [r for r in step if "sst_layer_bwd_kernel" in r["Kernel_Name"]]           # the one-launch encoder backward (round 5)
if lay_b:
    window("encoder backward ", lay_b[0], lay_b[-1])
else:
    window("encoder backward ", enc_att_b[0] if enc_att_b else None, enc_att_b[-1] if enc_att_b else None)
lay_f = [r for r in step if "sst_layer_fwd_kernel" in r["Kernel_Name"]]
if lay_f:
    window("encoder forward  ", lay_f[0], lay_f[-1])
window("whole step       ", step[0], step[-1])
