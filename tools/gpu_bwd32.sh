#!/bin/bash
# reverse neighbor pass on the 32x32x16 matrix shape (VSSR_EDGE_BWD_32=1): same-box A/B of bench.py over build/variants/lib_*.so
mkdir -p gpurun_out/bwd32
for rep in 1 2; do
for f in "" build/variants/lib_*.so; do
for v in 0 1; do
  [ "$v" = 0 ] && [ -n "$f" ] && continue
  VSSR_EVAL_LIB=${f:+$PWD/$f} VSSR_EDGE_BWD_32=$v timeout 300 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('${f:-product} bwd32=$v', 'evals/s %.0f' % d['value'], 'ms %.3f' % d['ms_per_step'], ' '.join('%s=%.3f' % (n[:12], x) for n, x in k.items()))" | tee -a gpurun_out/bwd32/ab.log
done; done; done
