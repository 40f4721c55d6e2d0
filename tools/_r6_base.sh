set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ROUNDS=2 bash tools/ab_env.sh "X=0" "GEOMAE_FUSED_MAX_TOKENS=200000" "GEOMAE_BUNDLE_CAP=32" > gpurun_out/r6_base_ab.txt 2>&1
BENCH_ARGS="--steps 15 --warmup 5 --no-cpu-baseline --workload nuscenes10" ROUNDS=1 bash tools/ab_env.sh "X=0" "GEOMAE_FUSED_MAX_TOKENS=2000000" >> gpurun_out/r6_base_ab.txt 2>&1
cat gpurun_out/r6_base_ab.txt
