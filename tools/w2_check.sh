cd /root/repo
GEOMAE_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/b_w2.json 2> gpurun_out/b_w2.err; tail -3 gpurun_out/b_w2.err
GEOMAE_FORCE_EXCHANGE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_fx.json 2> gpurun_out/b_fx.err; tail -2 gpurun_out/b_fx.err
