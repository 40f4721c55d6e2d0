set -x
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_eng_bench1.json 2> gpurun_out/r2_eng_bench1.err; tail -3 gpurun_out/r2_eng_bench1.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-engine --no-cpu-baseline > gpurun_out/r2_eng_bench_noeng.json 2>> gpurun_out/r2_eng_bench1.err
python - <<'PY'
import json
for f in ("gpurun_out/r2_eng_bench1.json", "gpurun_out/r2_eng_bench_noeng.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"], d["host_ms_per_step"], d.get("main_stream_phase_ms"), d.get("main_stream_phase_sum_ms"))
        print(json.dumps(d.get("roofline_hbm_kernels"), indent=1))
        print(d["roofline"])
    except Exception as e:
        print(f, "ERR", e)
PY
