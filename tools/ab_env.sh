#!/bin/bash
# same-box A/B of bench.py under different environment settings:  tools/ab_env.sh "VAR=a" "VAR=b OTHER=c" ...   (ROUNDS=2)
# prints ms_per_step + the main-stream phases per setting, alternating over ROUNDS rounds
ROUNDS=${ROUNDS:-2}
ARGS=${BENCH_ARGS:---steps 30 --warmup 8 --no-cpu-baseline}
for r in $(seq 1 $ROUNDS); do
  for setting in "$@"; do
    out=$(env $setting python bench.py $ARGS 2>/dev/null | tail -1)
    python - "$setting" "$out" <<'PY'
import json, sys
b = json.loads(sys.argv[2])
p = b.get("main_stream_phase_ms", {})
dw = ([v for k, v in [("roofline", b["roofline"])] if v and v.get("kernel") == "dw_kernel"] or [b["roofline_other_kernels"].get("dw_kernel")])[0]
print(f"{sys.argv[1]:40s} {b['ms_per_step']:.3f} ms  enc_fwd {p.get('enc_fwd')} dec_fwd {p.get('dec_fwd')} dec_bwd {p.get('dec_bwd')} enc_bwd {p.get('enc_bwd')} "
      f"vfe_bwd_join {p.get('vfe_bwd_join')}  dw/step {dw and dw['ms_per_step']}", flush=True)
PY
  done
done
