#!/bin/bash
for a in 0 7 8 24; do
  VSSR_ABLATE=$a timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('ablate=$a edge_fwd %.3f ms/step  (per launch %.3f)' % (k['edge_message_fwd'], d['roofline']['avg_launch_ms']))"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof/bench_kernel_stats.csv')):
    print(r['Name'][:70].ljust(72), r['Calls'].rjust(4), '%10.1f us avg' % (float(r['AverageNs'])/1e3))
PY
