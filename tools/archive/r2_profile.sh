#!/bin/bash
# Round-2 measurement artefacts for profiles/: kernel stats, per-dispatch trace of one step, PMC passes (HBM traffic, SQ issue /
# wait, MFMA busy), the default bench line and the two-ranks-on-one-GPU bench line.  Run on the GPU box: bash tools/r2_profile.sh
set -x
cd /root/repo
A="--steps 40 --warmup 10 --no-cpu-baseline"
S="--steps 3 --warmup 2 --no-cpu-baseline"
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver_cmd.json 2>> gpurun_out/r02_bench.err
bash tools/prof.sh r02 $A
bash tools/trace.sh r02 --steps 12 --warmup 6 --no-cpu-baseline
python tools/timeline_digest.py gpurun_out/trace_r02/kernel_trace.csv +8 > gpurun_out/r02_step_timeline.txt 2>&1
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|GRBM_GUI_ACTIVE\|SQ_BUSY_CU_CYCLES" | sort -u > /root/repo/gpurun_out/r02_mfma_counter_names.txt)
cat gpurun_out/r02_mfma_counter_names.txt
bash tools/pmc.sh r02_fetch FETCH_SIZE $S > /dev/null
bash tools/pmc.sh r02_write WRITE_SIZE $S > /dev/null
bash tools/pmc.sh r02_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" $S > /dev/null
bash tools/pmc.sh r02_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $S > /dev/null
ls -la gpurun_out/pmc_r02_*/
GEOMAE_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02_bench_2ranks_1gpu.json 2> gpurun_out/r02_bench_2ranks_1gpu.err
tail -c 600 gpurun_out/r02_bench_2ranks_1gpu.json; tail -3 gpurun_out/r02_bench_2ranks_1gpu.err
head -c 700 gpurun_out/r02_bench.json
