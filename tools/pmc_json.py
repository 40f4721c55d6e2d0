"""profiles/rNN_pmc_traffic.json from the two PMC summaries written by tools/pmc.sh (FETCH_SIZE and WRITE_SIZE passes).
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: counter unit KB, gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE reports half of a wide coalesced read stream)."""
import csv, json, sys
fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]
workload = sys.argv[4] if len(sys.argv) > 4 else "nuscenes1"
import re
def kernel_name(raw):
    """'void geomae::win_attn_fwd_kernel<4>(unsigned short const*; ...' -> 'win_attn_fwd_kernel' (template forms merged)"""
    return re.sub(r"<.*", "", raw.split("(")[0].replace("geomae::", "").replace("void ", "")).strip()
def load(path, col):
    acc = {}
    for r in csv.DictReader(open(path)):
        name = kernel_name(r["kernel"])
        v, n = float(r[col]), int(r["dispatches"])
        s, m = acc.get(name, (0.0, 0))
        acc[name] = (s + v * n, m + n)                     # dispatch-weighted mean over the forms of one kernel
    return {k: (s / max(m, 1), m) for k, (s, m) in acc.items()}
f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
res = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh, bench.py "
                   f"--workload {workload} --steps 3 --warmup 2, 4 frames per GPU, mean over all launches of the kernel). Counter unit KB; "
                   "corrected as MI355X_MICROARCH.md prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024. "
                   "Regenerate: tools/pmc.sh fetch2 FETCH_SIZE ...; tools/pmc.sh write2 WRITE_SIZE ...; tools/pmc_json.py",
       "_raw_kb": {}}
# the whole step: bytes of every dispatch of the run / steps of the run (adamw_kernel runs once per step)
steps = f.get("adamw_kernel", (0, 0))[1]
if steps:
    res["_bytes_per_step"] = int(sum((2 * f[k][0] + w[k][0]) * 1024 * f[k][1] for k in set(f) & set(w)) / steps)
    res["_steps_in_run"] = steps
for k in sorted(set(f) & set(w)):
    if not (k.startswith("sst_") or k.startswith("win_") or k.startswith("vfe_") or k.startswith("voxelize") or
            k.startswith("scan_") or k in ("dw_kernel", "dw_layer_kernel", "dw_layer_reduce_kernel", "sst_layer_fwd_kernel", "sst_layer_bwd_kernel", "sst_layer_fwd_big_kernel", "sst_stack_fwd_kernel", "dw_reduce_kernel", "heads_loss_kernel", "adamw_kernel", "hist_kernel", "place_kernel",
                                            "centroid_targets_kernel", "normal_curv_kernel", "normal_eig_kernel",
                                            "occ_count_kernel", "random_mask_kernel", "random_mask_win_kernel", "zero_arena_kernel", "grad_sumsq_kernel",
                                            "pack_weights_kernel", "rows_to_blocked_f32_kernel", "gather_token_coors_kernel")):
        continue
    res[k] = int((2 * f[k][0] + w[k][0]) * 1024)
    res["_raw_kb"][k] = [f[k][0], w[k][0], f[k][1]]
# the profiler id GEOMAE_KERNEL_DW (bench.py's "dw_kernel") times the old-form AND the layer-form contraction launches: their
# launch-weighted mean, under the name bench.py looks up
if "dw_layer_kernel" in res:
    names = [k for k in ("dw_kernel", "dw_layer_kernel") if k in res]
    tot = sum(res[k] * f[k][1] for k in names)
    cnt = sum(f[k][1] for k in names)
    res["_dw_kernel_old_form_only"] = res.get("dw_kernel")
    res["dw_kernel"] = int(tot / max(cnt, 1))
    res["_dw_kernel_note"] = "launch-weighted mean of dw_kernel (heads, VFE layer 1) and dw_layer_kernel (the stacks' layers, <= 4 per launch)"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
