#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   -- rocprofv3 kernel-trace stats of bench.py, summaries only
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/bench.py "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
mkdir -p /root/repo/gpurun_out/prof_$tag
# fixed names (rocprofv3 prefixes them with its pid: merged into a local gpurun_out/ an older run's file would linger)
find /tmp/prof_$tag -name '*kernel_stats.csv' -exec cp {} /root/repo/gpurun_out/prof_$tag/kernel_stats.csv \;
find /tmp/prof_$tag -name '*domain_stats.csv' -exec cp {} /root/repo/gpurun_out/prof_$tag/domain_stats.csv \;
ls -la /root/repo/gpurun_out/prof_$tag
