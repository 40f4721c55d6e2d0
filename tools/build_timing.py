"""Build tools/libgeomae_timing.so = the product sources with -DGEOMAE_PHASE_TIMING (clock64 stamps)."""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "geomae_amd", "csrc")
OBJ = os.path.join(ROOT, "tools", "build_timing")
os.makedirs(OBJ, exist_ok=True)
sys.path.insert(0, ROOT)
from geomae_amd.csrc.build import flags_for  # the product's per-file flags
objs, jobs = [], []
for src in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
    o = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    objs.append(o)
    stamp = [] if os.environ.get("GEOMAE_TIMING_NO_STAMPS") else ["-DGEOMAE_PHASE_TIMING"]     # (plain A/B builds of a macro)
    jobs.append(["/opt/rocm/bin/hipcc"] + flags_for(src) + stamp + os.environ.get("GEOMAE_TIMING_DEFS", "").split() + ["-c", src, "-o", o])
with ThreadPoolExecutor(8) as ex:
    list(ex.map(subprocess.check_call, jobs))
out = os.path.join(ROOT, "tools", os.environ.get("GEOMAE_TIMING_OUT", "libgeomae_timing.so"))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
