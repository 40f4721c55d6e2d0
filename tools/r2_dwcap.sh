for c in 0 12 16 20 24; do
  echo "cap $c"; GEOMAE_DW_CAP=$c python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['main_stream_phase_ms'])"
done
