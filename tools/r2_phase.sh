cd /root/repo
GEOMAE_TIMING_DEFS="-DGEOMAE_STAMP_MAX_GRID=300" python tools/build_timing.py > /dev/null 2>&1 && python tools/phase_timing.py 2>&1 | tail -24
echo ===== decoder size
GEOMAE_TIMING_DEFS="-DGEOMAE_STAMP_MIN_GRID=400" python tools/build_timing.py > /dev/null 2>&1 && python tools/phase_timing.py 2>&1 | tail -24
