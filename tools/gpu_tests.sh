#!/bin/bash
# full GPU test-suite + quick bench (iteration helper)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} 2>&1 | tail -40 > gpurun_out/pytest_gpu_full.log
tail -25 gpurun_out/pytest_gpu_full.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/quick_bench.log 2> gpurun_out/quick_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_bench.log').read().strip().splitlines()[-1])
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
tail -2 gpurun_out/quick_bench.err
