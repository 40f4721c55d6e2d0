for w in nuscenes10 waymo; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_$w.json 2> gpurun_out/r02_bench_$w.err
  python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['step_ms'], d['host_ms_per_step'], d['config']['workload']); print(d['main_stream_phase_ms']); print({k:(v['achieved'],v['avg_ms']) for k,v in d['roofline_hbm_kernels'].items()}); print(d['roofline'])"
  tail -2 gpurun_out/r02_bench_$w.err
done
python bench.py --frames-per-gpu 16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B16', d['value'], d['ms_per_step'], d['host_ms_per_step'])"
