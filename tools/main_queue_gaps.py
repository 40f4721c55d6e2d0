"""Idle gaps of the MAIN stream's hardware queue inside one step of a rocprofv3 kernel trace (tools/trace.sh): for every
kernel of the queue that holds adamw_kernel, the time since the previous kernel of that queue ended, when it exceeds a
threshold -- where the critical path waits for a side stream (or for the host).
usage: main_queue_gaps.py kernel_trace.csv [+step_index] [min_gap_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sel = sys.argv[2] if len(sys.argv) > 2 else "+8"
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
steps, prev = [], None
for i in ends:
    if prev is not None and i - prev > 5:
        steps.append((prev + 1, i))
    prev = i
lo, hi = steps[int(sel[1:])]
mainq = rows[hi]["Queue_Id"]
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("geomae::", "")[:40]
t0 = int(rows[lo - 1]["End_Timestamp"])          # end of the previous step's adamw
last_end, last_name, total = t0, "adamw_kernel (previous step)", 0.0
for r in rows[lo:hi + 1]:
    if r["Queue_Id"] != mainq:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end) / 1e3
    if gap > thr:
        total += gap
        print(f"{(s - t0) / 1e3:9.1f} us: {gap:6.1f} us idle before {name(r)} (after {last_name})")
    last_end, last_name = max(last_end, e), name(r)
print(f"# main queue q{mainq}: idle in gaps > {thr} us: {total:.1f} us of {(int(rows[hi]['End_Timestamp']) - t0) / 1e3:.1f} us")
