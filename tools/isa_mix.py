"""Instruction mix of every kernel in one HIP source (gfx950): how many VALU / MFMA / LDS / vector-memory / scalar
instructions the straight-line body holds.  With one wave per SIMD the layer kernels' duration is the issue time of
that stream, so this is the first thing to look at before touching one.
    python tools/isa_mix.py geomae_amd/csrc/sst_layer.hip"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include",
                               f"-I{ROOT}/geomae_amd/csrc", "-S", "--cuda-device-only", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        txt = open(out).read()
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
    print(f"{'kernel':44s} {'total':>6s} {'valu':>6s} {'mfma':>5s} {'lds':>5s} {'vmem':>5s} {'salu':>5s} {'wait':>5s}")
    for i in range(1, len(parts), 2):
        body = parts[i + 1].split(".Lfunc_end")[0]
        c = collections.Counter()
        for line in body.splitlines():
            t = line.strip()
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            op = t.split()[0]
            kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else
                    "wait" if op.startswith(("s_waitcnt", "s_nop", "s_barrier")) else "salu" if op.startswith("s_") else
                    "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_")) else "other")
            c[kind] += 1
            c["total"] += 1
        m = re.match(r"_ZN6geomae(\d+)", parts[i])                       # geomae::<name>: length-prefixed in the mangling
        name = parts[i][len(m.group(0)):][:int(m.group(1))] if m else parts[i]
        print(f"{name[:44]:44s} {c['total']:6d} {c['valu']:6d} {c['mfma']:5d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:5d} {c['wait']:5d}")


if __name__ == "__main__":
    main(sys.argv[1])
