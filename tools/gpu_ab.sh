#!/bin/bash
# A/B helper: bench every build/variants/lib_*.so on the same box, AB_REPS times (default 2), print the per-kernel split.
# The variant is selected through VSSR_EVAL_LIB; the product library is never overwritten.
mkdir -p gpurun_out
for rep in $(seq 1 ${AB_REPS:-2}); do
for f in build/variants/lib_*.so; do
  VSSR_EVAL_LIB=$PWD/$f python bench.py --steps ${AB_STEPS:-8} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$f', 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/ab.log
done; done
