cd /root/repo
bash tools/trace.sh gaps --steps 12 --warmup 6 --no-cpu-baseline --profile-every 100 > /dev/null
for s in +7 +8 +9; do python tools/main_queue_gaps.py gpurun_out/trace_gaps/kernel_trace.csv $s 3 ; done > gpurun_out/main_gaps.txt 2>&1
python tools/timeline_digest.py gpurun_out/trace_gaps/kernel_trace.csv +8 > gpurun_out/tl_now.txt 2>&1
rm -rf gpurun_out/trace_gaps
