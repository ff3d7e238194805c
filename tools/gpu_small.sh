#!/bin/bash
# per-atom cost of SMALL chains (the reference's own 60-atom 2x2 slab + adsorbates) at several batch sizes
mkdir -p gpurun_out
for cfg in "80 256" "80 1024" "80 4096" "140 1024" "260 256"; do
  set -- $cfg
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --atoms-per-chain $1 --chains-per-gpu $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('atoms/chain $1 chains $2', 'atoms', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/small.log
done
