#!/bin/bash
# per-atom cost of larger slabs (secondary bench lines): 260 (16-feature slices) vs 380 / 480 (8-feature slices)
mkdir -p gpurun_out
for n in 260 380 480; do
  python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --atoms-per-chain $n --chains-per-gpu ${CHAINS:-256} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('atoms/chain $n', 'atoms', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/sizes.log
done
