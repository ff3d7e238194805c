cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_relax
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_relax -o relax -- python3 tools/bench_relax.py --relax-steps 20 > gpurun_out/prof_relax.log 2>&1
f=$(find gpurun_out/prof_relax -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print(r['Name'].split('(')[0][:56].ljust(58), r['Calls'].rjust(5), '%9.1f us avg' % (float(r['AverageNs'])/1e3), '%6.2f%%' % (100*float(r['TotalDurationNs'])/tot))
PY
find gpurun_out/prof_relax -name "*kernel_trace.csv" -delete
