python -m pytest tests -m gpu -q -x 2>&1 | tail -8
for cfg in "1 0" "1 32" "1 48" "1 64" "1 144" "0 0" "0 144"; do set -- $cfg; 
  GEOMAE_FUSED_LAYERS=$1 GEOMAE_BUNDLE_CAP=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['main_stream_phase_ms']
print('fused=$1 cap=$2', d['ms_per_step'], {k:p[k] for k in ('enc_fwd','dec_fwd','dec_bwd','enc_bwd')})"
done
