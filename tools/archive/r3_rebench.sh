cd /root/repo
for wl in nuscenes1 nuscenes10 waymo; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b_$wl.json 2> /tmp/b_$wl.err; cp /tmp/b_$wl.json gpurun_out/r03_bench_$wl.json
done
python bench.py > /tmp/b.json 2> /tmp/b.err; cp /tmp/b.json gpurun_out/r03_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b.json 2>> /tmp/b.err; cp /tmp/b.json gpurun_out/r03_bench_driver_cmd.json
GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b.json 2>> /tmp/b.err; cp /tmp/b.json gpurun_out/r03_bench_nccl_w1.json
