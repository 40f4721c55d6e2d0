#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> "<counters>" <command...>  -- rocprofv3 PMC pass (kernel-trace only) of any command,
# per-kernel averages; PMC_FILTER=<substring> keeps matching kernels only
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -- "$@" > /root/repo/gpurun_out/pmc_$tag.log 2>&1
mkdir -p /root/repo/gpurun_out/pmc_$tag
python - <<PY
import csv, glob, collections, os
files = glob.glob('/tmp/pmc_$tag/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
flt = os.environ.get('PMC_FILTER', '')
for f in files:
    for r in csv.DictReader(open(f)):
        if flt and flt not in r['Kernel_Name']: continue
        k = r['Kernel_Name'][:48] + ' g' + r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
with open('/root/repo/gpurun_out/pmc_$tag/summary.csv', 'w') as out:
    names = sorted({c for v in agg.values() for c in v})
    out.write('kernel,dispatches,' + ','.join(names) + '\n')
    for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
        n = max(cnt[(k, c)] for c in names if (k, c) in cnt)
        out.write(k.replace(',', ';') + f',{n},' + ','.join(f'{v.get(c, 0)/n:.1f}' for c in names) + '\n')
print(open('/root/repo/gpurun_out/pmc_$tag/summary.csv').read()[:3000])
PY
