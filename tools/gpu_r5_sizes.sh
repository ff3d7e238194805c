#!/bin/bash
# round 5: per-atom cost by chain size with the final defaults (256 chains like profiles/r04/bench_chain_sizes.txt; 128 chains for 1 400)
O=gpurun_out/r5_sizes; mkdir -p $O
for atoms in 260 380 480 700 1000; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --atoms-per-chain $atoms 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('atoms/chain $atoms atoms', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes.txt
done
for atoms in 260 700 1000 1400; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --chains-per-gpu 128 --atoms-per-chain $atoms 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('128 chains, atoms/chain $atoms atoms', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes.txt
done
bash tools/gpu_suite.sh
