"""Register / scratch budget of the hot layer kernels: the documents say "no scratch" for the one-launch
layer kernels -- this test compiles the two sources with the product's flags and -Rpass-analysis=kernel-resource-usage and fails
when a kernel's scratch or spill count leaves the budget written here (DESIGN.md section 3 quotes the same numbers), so the claim
cannot drift again.  hipcc cross-compiles without a GPU; ~25 s."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "geomae_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel (substring of the mangled name) -> (max scratch bytes per lane, max spilled VGPRs, max LDS bytes per workgroup)
BUDGET = {
    "sst_fused.hip": {
        "20sst_layer_fwd_kernel": (0, 0, 160 * 1024),
        "20sst_layer_bwd_kernel": (0, 0, 160 * 1024),
    },
    "sst_ws.hip": {
        "sst_layer_fwd_ws_kernelILb0E": (0, 0, 160 * 1024),
        "sst_layer_fwd_ws_kernelILb1E": (0, 0, 160 * 1024),
    },
}


def _report(src):
    import sys
    sys.path.insert(0, ROOT)
    from geomae_amd.csrc.build import flags_for
    path = os.path.join(SRC, src)
    r = subprocess.run([HIPCC] + flags_for(path) + ["-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = block.split()[0]
        val = lambda pat: int(re.search(pat, block).group(1))
        out[name] = dict(vgpr=val(r"    VGPRs: (\d+)"), scratch=val(r"ScratchSize \[bytes/lane\]: (\d+)"),
                         spill=val(r"VGPRs Spill: (\d+)"), lds=val(r"LDS Size \[bytes/block\]: (\d+)"))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_layer_kernels_stay_inside_their_register_budget():
    with ThreadPoolExecutor(2) as ex:
        reports = dict(zip(BUDGET, ex.map(_report, BUDGET)))
    seen = []
    for src, kernels in BUDGET.items():
        for key, (max_scratch, max_spill, max_lds) in kernels.items():
            match = [(n, v) for n, v in reports[src].items() if key in n]
            assert len(match) == 1, (src, key, list(reports[src]))
            name, v = match[0]
            seen.append((name, v))
            assert v["scratch"] <= max_scratch and v["spill"] <= max_spill and v["lds"] <= max_lds and v["vgpr"] <= 256, (name, v)
    print(seen)
