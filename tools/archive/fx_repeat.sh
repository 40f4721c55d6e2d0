cd /root/repo
for q in "" 8; do for i in 1 2 3 4 5 6; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-every 1000 2>/dev/null > /tmp/o.json; python - "$q" <<'PY'
import json,sys
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[0]); p=d["main_stream_phase_ms"]
print("queues", sys.argv[1] or "default", d["ms_per_step"], "dec_fwd", p["dec_fwd"], "dec_bwd", p["dec_bwd"], "enc_bwd", p["enc_bwd"], d["stream_probe"]["pairs_tried"])
PY
done; done
unset GPU_MAX_HW_QUEUES
for q in "" 8; do for i in 1 2 3; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-every 1000 2>/dev/null > /tmp/o.json; python - "$q" <<'PY'
import json,sys
d=json.loads(open("/tmp/o.json").read().strip().splitlines()[0]); p=d["main_stream_phase_ms"]
print("plain queues", sys.argv[1] or "default", d["ms_per_step"], "dec_fwd", p["dec_fwd"], "dec_bwd", p["dec_bwd"])
PY
done; done
