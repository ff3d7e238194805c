#!/bin/bash
# The round's evidence run: whole GPU suite, headline bench (two / one engine), rocprofv3 kernel stats of both commands, PMC
# passes (tools/gpu_pmc.sh), the MC loop and the relaxation driver.  Everything lands in gpurun_out/final/; copy what is to be
# judged to profiles/rNN/.
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error|^FAILED|^ERROR" $O/pytest_gpu.log | tail -5
grep -A3 "deviation from the fp64 oracle" $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python3 -c "
import json; d = json.load(open('$O/bench_n1.json')); print('bench', round(d['value'], 1), round(d['ms_per_step'], 3), d['kernel_ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['executed_pipe']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('gflops_per_core'))"
python bench.py --streams 1 --no-cpu-baseline > $O/bench_n1_streams1.json 2>> $O/bench_n1.err
for s in 1 2; do
  rm -rf $O/prof_s$s
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s$s -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams $s > $O/bench_under_rocprof_streams$s.json 2> $O/rocprof_s$s.err
  f=$(find $O/prof_s$s -name '*kernel_stats.csv' | head -1); cp "$f" $O/rocprof_kernel_stats_streams$s.csv; head -8 $O/rocprof_kernel_stats_streams$s.csv
  rm -rf $O/prof_s$s
done
bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; tail -12 $O/pmc.log
cp gpurun_out/pmc/pmc_summary.json gpurun_out/pmc/pmc_summary.txt $O/ 2>/dev/null
python tools/bench_mc.py > $O/bench_mc.json 2>/dev/null; cut -c1-400 $O/bench_mc.json
python tools/bench_relax.py > $O/bench_relax.jsonl 2>/dev/null; cut -c1-300 $O/bench_relax.jsonl
