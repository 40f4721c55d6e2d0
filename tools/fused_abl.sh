#!/bin/bash
# timing-library ablations of the one-launch layer kernel: bash tools/fused_abl.sh "<defs>" ...   (run on the GPU box)
cd /root/repo
for defs in "$@"; do
  GEOMAE_TIMING_DEFS="$defs" python tools/build_timing.py > /dev/null 2>&1
  echo "=== defs: $defs"
  python tools/fused_layer_time.py enc 2>&1 | grep -E "one-launch=1|wave 0|A:|B:|C:|D:|LN1|E:|F:|LN2" | head -12
done
