#!/bin/bash
# A/B helper: bench every gpu_variants/lib_*.so on the same box, twice, print the per-kernel split
cp surface-sampling_amd/libvssr_eval.so /tmp/lib_keep.so
for rep in 1 2; do
for f in gpu_variants/lib_*.so; do
  cp $f surface-sampling_amd/libvssr_eval.so
  python bench.py --steps ${AB_STEPS:-8} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$f', 'evals/s %.0f' % d['value'], ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))"
done; done
cp /tmp/lib_keep.so surface-sampling_amd/libvssr_eval.so
