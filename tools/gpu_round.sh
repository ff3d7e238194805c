#!/bin/bash
# End-of-milestone evidence run: GPU tests, the default bench line, rocprofv3 kernel stats of the same command, PMC traffic of
# the edge kernels.  Everything lands in gpurun_out/ (copy what should be judged into profiles/rNN/).
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value %.0f  ms/step %.3f  roofline.frac %.3f  pcie %.0f  cpu %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'],
      d['pcie_inclusive']['value'] if d['pcie_inclusive'] else -1, d['cpu_baseline'] and round(d['cpu_baseline']['value'], 2)))
print({k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/prof.err
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats.csv && head -12 "$f" | cut -c1-150
find gpurun_out/prof -name "*kernel_trace.csv" -delete
bash tools/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; tail -25 gpurun_out/traffic.log
