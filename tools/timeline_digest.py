"""Digest of one training step from a rocprofv3 kernel trace (tools/trace.sh): start (us), duration, gap to the
latest previous end over all queues, queue, workgroups, kernel.  usage: timeline_digest.py kernel_trace.csv [step_index_from_end | +step_index_from_start]
(bench.py's last steps are its untimed extra passes, whose batches are not prefetched: pick a step of the timed region,
e.g. +10 with --warmup 6)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sel = sys.argv[2] if len(sys.argv) > 2 else "2"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
# a step = after the last adamw launch of step k-1 ... last adamw launch of step k (two launches per step: segments)
steps = []
prev = None
for i in ends:
    if prev is not None and i - prev > 5:
        steps.append((prev + 1, i))
    prev = i
lo, hi = steps[int(sel[1:])] if sel.startswith("+") else steps[-int(sel)]
t0 = int(rows[lo]["Start_Timestamp"])
last_end = t0
busy = 0.0
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("geomae::", "").replace("void at::native::", "at::")[:44]
    wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    gap = (s - last_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f} q{r['Queue_Id']} {wgs:6d} {name}")
    if e > last_end:
        busy += (e - max(s, last_end)) / 1e3
        last_end = e
print(f"# step span {(last_end - t0) / 1e3:.1f} us, union-busy {busy:.1f} us, launches {hi - lo + 1}")
