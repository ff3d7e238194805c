#!/bin/bash
# A/B of build/variants/lib_*.so on the 480-atom secondary workload
mkdir -p gpurun_out
for rep in 1 2; do
for f in build/variants/lib_*.so; do
  VSSR_EVAL_LIB=$PWD/$f python bench.py --steps 6 --warmup 2 --no-cpu-baseline --atoms-per-chain ${ATOMS:-480} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$f', 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/ab480.log
done; done
