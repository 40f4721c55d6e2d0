# forced-exchange (RCCL, one rank) step time under different runtime settings; usage: bash tools/fx_envs.sh
run() { tag=$1; shift; env "$@" GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/o.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[0])
p=d["main_stream_phase_ms"]
print("$tag", d["ms_per_step"], "dec_fwd", p["dec_fwd"], "dec_bwd", p["dec_bwd"], "vfe_fwd", p["vfe_fwd"], "opt", p["optimizer"], d.get("stream_probe"))
PY
}
for q in 4 5 6 7 8; do
run high_q$q GPU_MAX_HW_QUEUES=$q
run normal_q$q GPU_MAX_HW_QUEUES=$q GEOMAE_COMM_PRIORITY=normal
run normal_noprobe_q$q GPU_MAX_HW_QUEUES=$q GEOMAE_COMM_PRIORITY=normal GEOMAE_STREAM_PROBE=0
done
