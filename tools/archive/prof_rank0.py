"""Run bench.py under cProfile on rank 0 (other ranks run it plainly): where does the host time of an N > 1 step go?
usage: python -m torch.distributed.run --nproc-per-node 2 ... tools/prof_rank0.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline"""
import cProfile
import os
import pstats
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
if os.environ.get("RANK", "0") == "0":
    pr = cProfile.Profile()
    pr.enable()
    try:
        runpy.run_path(sys.argv[0], run_name="__main__")
    finally:
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
else:
    runpy.run_path(sys.argv[0], run_name="__main__")
