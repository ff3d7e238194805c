#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for s in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams $s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $s', 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'roofline frac %.3f' % d['roofline']['frac'])" | tee -a gpurun_out/streams.log
done; done
