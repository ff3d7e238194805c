#!/bin/bash
# samples rocm-smi (clocks, power) while the bench runs: is the device at its power limit under this workload?
mkdir -p gpurun_out
(for i in $(seq 1 12); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor junction\)" | tr '\n' ' '; echo; sleep 1; done) > gpurun_out/clocks.txt 2>&1 &
SM=$!
python bench.py --steps 600 --warmup 20 --no-cpu-baseline > gpurun_out/clocks_bench.log 2>&1
wait $SM
cat gpurun_out/clocks.txt | head -14
python - <<'PY'
import json
d=json.loads(open('gpurun_out/clocks_bench.log').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])
PY
