#!/bin/bash
# alternating same-box A/B of the step with the one-launch layer forward on (1) / off (0): bash tools/ab_fused.sh [rounds]
cd /root/repo
for r in $(seq 1 ${1:-3}); do for m in 1 0; do
  GEOMAE_FUSED_LAYERS=$m python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); p=d['main_stream_phase_ms']
print('fused=$m', d['ms_per_step'], d['step_ms']['p50'], {k:p[k] for k in ('enc_fwd','dec_fwd','dec_bwd','enc_bwd')})"
done; done
