"""Register / LDS / occupancy table of every kernel of the listed csrc files (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py [file stems ...]   (default: the layer, attention, VFE, heads and fused-layer sources)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "geomae_amd", "csrc")
stems = sys.argv[1:] or ["sst_layer", "window", "vfe", "heads_loss", "sst_fused"]
KEYS = [("VGPR", r"    VGPRs: (\d+)"), ("AGPR", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
        ("LDS", r"LDS Size \[bytes/block\]: (\d+)")]
os.makedirs("/tmp/kres", exist_ok=True)
for f in stems:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
                        "-ffp-contract=fast", "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage",
                        "-c", os.path.join(SRC, f + ".hip"), "-o", f"/tmp/kres/{f}.o"], capture_output=True, text=True)
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split("\n")[0].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        name = re.sub(r"\(.*", "", name).replace("geomae::", "").replace("void ", "")
        vals = []
        for k, pat in KEYS:
            m = re.search(pat, b)
            vals.append(f"{k} {m.group(1) if m else '?':>6s}")
        print(f"{f:10s} {name[:40]:40s} " + "  ".join(vals))
