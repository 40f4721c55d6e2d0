#!/bin/bash
# usage: tools/trace.sh <tag> <bench args...>   -- rocprofv3 per-dispatch kernel trace of bench.py (csv)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python /root/repo/bench.py "$@" > /root/repo/gpurun_out/trace_$tag.log 2>&1
mkdir -p /root/repo/gpurun_out/trace_$tag
find /tmp/trace_$tag -name '*kernel_trace.csv' -exec cp {} /root/repo/gpurun_out/trace_$tag/kernel_trace.csv \;
ls -la /root/repo/gpurun_out/trace_$tag
