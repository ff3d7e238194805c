#!/bin/bash
# A/B of one environment knob on the same box: tools/gpu_ab_env.sh VAR v1 v2 ...   (bench steps: AB_STEPS, repetitions: AB_REPS)
mkdir -p gpurun_out
var=$1; shift
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in "$@"; do
  env $var=$v python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$var=$v', 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/ab_env.log
done; done
