#!/bin/bash
O=gpurun_out/r5_i; mkdir -p $O
for rep in 1 2; do for flag in 0 1; do for atoms in 480 700; do
  VSSR_EDGE_FWD_MPASS_FS8=$flag python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --chains-per-gpu 128 --atoms-per-chain $atoms 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('fwd_mpass_fs8=$flag atoms/chain $atoms atoms', a, 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/ab_fwd_mpass_fs8.txt
done; done; done
