"""profiles/rNN_mfma_util.json from the MFMA PMC pass (tools/pmc.sh ... "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
SQ_INSTS_VALU_MFMA_MOPS_F32 ...") and the rocprofv3 kernel stats of the same build:
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES per dispatch (summed over the chip's SIMDs) / (1024 SIMDs x average duration x 2.4 GHz).
usage: mfma_json.py <pmc summary.csv> <kernel_stats.csv> <out.json>"""
import csv, json, re, sys


def kernel_name(raw):        # as tools/pmc_json.py
    return re.sub(r"<.*", "", raw.split("(")[0].replace("geomae::", "").replace("void ", "")).strip()


pmc, stats, out = sys.argv[1:4]
dur = {}
for r in csv.DictReader(open(stats)):
    n = kernel_name(r["Name"])
    t, c = dur.get(n, (0.0, 0))
    dur[n] = (t + float(r["TotalDurationNs"]), c + int(r["Calls"]))
acc = {}
for r in csv.DictReader(open(pmc)):
    n, d = kernel_name(r["kernel"]), int(r["dispatches"])
    a = acc.setdefault(n, dict(busy=0.0, bf16=0.0, f32=0.0, n=0))
    a["busy"] += float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) * d
    a["bf16"] += float(r["SQ_INSTS_VALU_MFMA_MOPS_BF16"]) * d
    a["f32"] += float(r["SQ_INSTS_VALU_MFMA_MOPS_F32"]) * d
    a["n"] += d
res = {"_comment": "MFMA busy fraction per kernel = SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc pass of tools/r2_profile.sh, mean per "
                   "dispatch, summed over the chip's SIMDs) / (1024 SIMDs x average kernel duration from the kernel stats of the "
                   "same build x 2.4 GHz); mops_bf16 = SQ_INSTS_VALU_MFMA_MOPS_BF16 per dispatch; template forms of one kernel merged"}
for n, a in sorted(acc.items(), key=lambda kv: -kv[1]["busy"]):
    if a["busy"] <= 0 or n not in dur:
        continue
    us = dur[n][0] / dur[n][1] / 1e3
    res[n] = {"mfma_busy_frac": round(a["busy"] / a["n"] / (1024 * us * 2400.0), 4), "mfma_busy_cycles": round(a["busy"] / a["n"], 1),
              "avg_kernel_us": round(us, 2), "mops_bf16": round(a["bf16"] / a["n"], 1), "mops_f32": round(a["f32"] / a["n"], 1)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v["mfma_busy_frac"] for k, v in res.items() if not k.startswith("_")}, indent=1))
