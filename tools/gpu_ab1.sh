#!/bin/bash
# same-box A/B of build/variants/lib_*.so with ONE engine per GPU (clean per-kernel times); AB_REPS repetitions
mkdir -p gpurun_out
for rep in $(seq 1 ${AB_REPS:-2}); do
for f in build/variants/lib_*.so; do
  VSSR_EVAL_LIB=$PWD/$f python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --streams ${AB_STREAMS:-1} ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$f', 'evals/s %.0f' % d['value'], 'ms %.3f' % d['ms_per_step'], ' '.join('%s=%.3f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/ab1.log
done; done
