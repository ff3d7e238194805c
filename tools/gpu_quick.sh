#!/bin/bash
# quick perf + correctness iteration on the GPU box
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${QUICK_K:-intermediates or batched or determinism}" 2>&1 | tail -15 > gpurun_out/quick_pytest.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --no-cpu-baseline > gpurun_out/quick_bench.log 2> gpurun_out/quick_bench.err
tail -4 gpurun_out/quick_pytest.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_bench.log').read().strip().splitlines()[-1])
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'roofline', d['roofline']['achieved'], d['roofline']['avg_launch_ms'])
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
tail -2 gpurun_out/quick_bench.err
