#!/bin/bash
# Round-6 measurement artefacts for profiles/, PER WORKLOAD (nuscenes1 = BASELINE config 2, nuscenes10 = config 3's per-GPU
# load, waymo = config 4's geometry): rocprofv3 kernel stats, PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate
# passes; MFMA busy), the bench line.  For nuscenes1 also the per-dispatch timeline of one step, the SQ wait counters, the
# driver's command line and the forced one-rank RCCL exchange.  Run on the GPU box:  bash tools/r6_profile.sh [workloads...]
set -x
cd ${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
mkdir -p gpurun_out
WLS=${@:-nuscenes1 nuscenes10 waymo}
for wl in $WLS; do
  W="--workload $wl"
  A="$W --steps 40 --warmup 10 --no-cpu-baseline"
  S="$W --steps 3 --warmup 2 --no-cpu-baseline"
  python bench.py $W --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b_$wl.json 2> /tmp/b_$wl.err; cp /tmp/b_$wl.json gpurun_out/r06_bench_$wl.json
  bash tools/prof.sh r06_$wl $A > /dev/null
  cp gpurun_out/prof_r06_$wl/kernel_stats.csv gpurun_out/r06_${wl}_kernel_stats.csv
  bash tools/pmc.sh r06_${wl}_fetch FETCH_SIZE $S > /dev/null
  bash tools/pmc.sh r06_${wl}_write WRITE_SIZE $S > /dev/null
  bash tools/pmc.sh r06_${wl}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $S > /dev/null
  cp gpurun_out/pmc_r06_${wl}_fetch/summary.csv gpurun_out/r06_${wl}_pmc_fetch_size.csv
  cp gpurun_out/pmc_r06_${wl}_write/summary.csv gpurun_out/r06_${wl}_pmc_write_size.csv
  cp gpurun_out/pmc_r06_${wl}_mfma/summary.csv gpurun_out/r06_${wl}_pmc_mfma.csv
  python tools/pmc_json.py gpurun_out/r06_${wl}_pmc_fetch_size.csv gpurun_out/r06_${wl}_pmc_write_size.csv gpurun_out/r06_${wl}_pmc_traffic.json "$wl" > /dev/null
  python tools/mfma_json.py gpurun_out/r06_${wl}_pmc_mfma.csv gpurun_out/r06_${wl}_kernel_stats.csv gpurun_out/r06_${wl}_mfma_util.json > /dev/null
done
if echo "$WLS" | grep -q nuscenes1; then
  python bench.py > /tmp/b.json 2> /tmp/b.err; cp /tmp/b.json gpurun_out/r06_bench.json
  python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b.json 2>> /tmp/b.err; cp /tmp/b.json gpurun_out/r06_bench_driver_cmd.json
  GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> /tmp/b.err | grep "^{" | tail -1 > gpurun_out/r06_bench_nccl_w1.json
  bash tools/trace.sh r06 --steps 12 --warmup 6 --no-cpu-baseline > /dev/null
  python tools/timeline_digest.py gpurun_out/trace_r06/kernel_trace.csv +8 > gpurun_out/r06_step_timeline.txt 2>&1
  # HBM bytes per dispatch by kernel AND grid size (encoder- vs decoder-size launches), then the decoder-backward window of a
  # traced step: bytes of the dispatches that ran inside it / its duration
  PMC_BY_GRID=1 bash tools/pmc.sh r06_fetch_g FETCH_SIZE --steps 3 --warmup 2 --no-cpu-baseline > /dev/null
  PMC_BY_GRID=1 bash tools/pmc.sh r06_write_g WRITE_SIZE --steps 3 --warmup 2 --no-cpu-baseline > /dev/null
  python tools/window_bandwidth.py gpurun_out/trace_r06/kernel_trace.csv gpurun_out/pmc_r06_fetch_g/summary.csv gpurun_out/pmc_r06_write_g/summary.csv +8 > gpurun_out/r06_window_bandwidth.txt 2>&1
  rm -rf gpurun_out/trace_r06
  bash tools/pmc.sh r06_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" --steps 3 --warmup 2 --no-cpu-baseline > /dev/null
  cp gpurun_out/pmc_r06_sq/summary.csv gpurun_out/r06_pmc_sq_counters.csv
fi
# round 6: the contraction ALONE on the chip (bench.py's roofline.alone), the looping one-launch layer against the three-launch form
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I geomae_amd/csrc tools/dw_bench.hip -o tools/dw_bench && tools/dw_bench 21967 > gpurun_out/r06_microbench_dw_alone.txt 2>&1
{ python tools/ws_layer_time.py dec 64,96,144; SWEEPS=10 python tools/ws_layer_time.py dec 96,144; SWEEPS=10 python tools/ws_layer_time.py enc 64,96; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_ws_layer_alone.txt
if [ -f tools/libgeomae_timing.so ]; then
  { SWEEPS=10 python tools/ws_phase_time.py dec 144; python tools/ws_phase_time.py dec 64; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_ws_layer_phases.txt
fi
ROUNDS=2 bash tools/ab_env.sh "GEOMAE_WS_LAYERS=0" "GEOMAE_WS_LAYERS=1" > gpurun_out/r06_ws_layer_in_step.txt 2>&1
BENCH_ARGS="--workload nuscenes10 --steps 12 --warmup 4 --no-cpu-baseline" ROUNDS=1 bash tools/ab_env.sh "GEOMAE_WS_LAYERS=0" "GEOMAE_WS_LAYERS=1" >> gpurun_out/r06_ws_layer_in_step.txt 2>&1
ls gpurun_out | grep r06_
