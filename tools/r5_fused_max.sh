for w in nuscenes10 waymo; do for mt in 12288 32768; do echo "== $w fused_max_tokens=$mt"; GEOMAE_FUSED_MAX_TOKENS=$mt python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readlines()[-1]); p=b[\"main_stream_phase_ms\"]; print(b[\"ms_per_step\"], b[\"value\"], {k:p[k] for k in (\"enc_fwd\",\"dec_fwd\",\"dec_bwd\",\"enc_bwd\",\"vfe_bwd_join\")})"; done; done
