#!/bin/bash
cd /root/repo
for defs in "$@"; do
  GEOMAE_TIMING_NO_STAMPS=1 GEOMAE_TIMING_DEFS="$defs" python tools/build_timing.py > /dev/null 2>&1
  echo "=== defs: $defs"
  LIB=tools/libgeomae_timing.so python tools/attn_time.py 2>&1 | grep shift
done
