#!/bin/bash
# full GPU test-suite + a short bench without the CPU baseline; logs under gpurun_out/
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/quick_pytest.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/quick_bench.log 2> gpurun_out/quick_bench.err
tail -6 gpurun_out/quick_pytest.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_bench.log').read().strip().splitlines()[-1])
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'roofline', d['roofline'].get('kernel'), round(d['roofline']['achieved'],1), d['roofline']['avg_launch_ms'])
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
tail -2 gpurun_out/quick_bench.err
