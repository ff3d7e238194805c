#!/bin/bash
# per-atom cost by chain size (secondary lines of DESIGN.md section 5.3): 256 chains of 260 .. 1 000 atoms, 128 chains up to 1 400
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
O=gpurun_out/sizes; mkdir -p $O; rm -f $O/bench_chain_sizes.txt
line () { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('$1', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes.txt; }
for atoms in 260 380 480 700 1000; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --atoms-per-chain $atoms 2>/dev/null | line "atoms/chain $atoms atoms"
done
for atoms in 260 700 1000 1400; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --chains-per-gpu 128 --atoms-per-chain $atoms 2>/dev/null | line "128 chains, atoms/chain $atoms atoms"
done
