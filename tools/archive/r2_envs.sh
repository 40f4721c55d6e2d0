run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['host_ms_per_step'].get('busy'), d['main_stream_phase_sum_ms'])"; }
run X=1
run GEOMAE_ENGINE_SERIAL=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=3
run GPU_MAX_HW_QUEUES=8
run HIP_FORCE_DEV_KERNARG=0
run AMD_DIRECT_DISPATCH=0
run HSA_ENABLE_SDMA=0
