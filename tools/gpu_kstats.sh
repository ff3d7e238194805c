#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/kstats_bench.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof/bench_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
    print(r['Name'].split('(')[0][:52].ljust(54), r['Calls'].rjust(4), '%9.1f us avg' % (float(r['AverageNs'])/1e3), '%6.2f%%' % (100*float(r['TotalDurationNs'])/tot))
PY
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/kstats.csv
