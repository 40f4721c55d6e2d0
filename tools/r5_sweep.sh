for w in nuscenes10 nuscenes1; do for ev in 2 4; do for g in 6 8 10; do echo "== $w every=$ev chunks=$g"; GEOMAE_ENC_DW_EVERY=$ev GEOMAE_DW_CHUNKS=$g python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readlines()[-1]); p=b[\"main_stream_phase_ms\"]; print(b[\"ms_per_step\"], b[\"value\"], {k:p[k] for k in (\"dec_bwd\",\"enc_bwd\",\"vfe_bwd_stats\",\"vfe_bwd_layer1\",\"vfe_bwd_join\")})"; done; done; done
