set -x
nproc; lscpu | head -20; cat /sys/fs/cgroup/cpu.max 2>/dev/null; env | grep -i -E "hip|hsa|rocm|gpu_|amd" | head -20
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_base_bench1.json 2> gpurun_out/r2_base_bench1.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_base_bench2.json 2>> gpurun_out/r2_base_bench1.err
python tools/host_time.py > gpurun_out/r2_base_host.txt 2>&1
taskset -c 0 python tools/host_time.py > gpurun_out/r2_base_host_1core.txt 2>&1
taskset -c 0,1 python tools/host_time.py > gpurun_out/r2_base_host_2core.txt 2>&1
cat gpurun_out/r2_base_bench1.json | cut -c1-300; cat gpurun_out/r2_base_bench2.json | cut -c1-300; cat gpurun_out/r2_base_host*.txt
