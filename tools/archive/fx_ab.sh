# A/B of an environment switch under the forced one-rank RCCL exchange; usage: bash tools/fx_ab.sh VAR v0 v1 [steps]
VAR=$1; A=$2; B=$3; N=${4:-60}
for rep in 1 2 3; do for v in $A $B; do
  env $VAR=$v GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps $N --warmup 10 --no-cpu-baseline --profile-every 1000 2>/dev/null > /tmp/o.json
  python - <<PY
import json
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[0])
p=d["main_stream_phase_ms"]
print("$VAR=$v", d["ms_per_step"], d["step_ms"], "phase sum", d["main_stream_phase_sum_ms"], "vfe_fwd", p["vfe_fwd"], "opt", p["optimizer"])
PY
done; done
