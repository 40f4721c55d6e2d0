#!/bin/bash
# only the bench lines of profiles/r04_bench*.json again (run AFTER the PMC / kernel-stats files of r4_profile.sh have been
# copied to profiles/: the instrumented kernel, roofline.traffic and step_roofline.hbm are read from there)
cd /root/repo
for wl in nuscenes1 nuscenes10 waymo; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_$wl.json 2> /tmp/b_$wl.err
done
python bench.py > gpurun_out/r04_bench.json 2> /tmp/b.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd.json 2>> /tmp/b.err
GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_nccl_w1.json 2>> /tmp/b.err
GEOMAE_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_2ranks_1gpu.json 2>> /tmp/b.err
tail -3 /tmp/b.err
