#!/bin/bash
# wide reverse workgroups (12 / 16 waves) for the 8- / 4-feature slices: same-box A/B through VSSR_EDGE_BWD_WIDE
O=gpurun_out/r04_fs4; mkdir -p $O
for rep in 1 2; do for w in 0 1; do for n in 480 700 1000 1400; do
  VSSR_EDGE_BWD_WIDE=$w python bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 --atoms-per-chain $n --chains-per-gpu 128 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('wide=$w atoms/chain $n', 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), 'edge_fwd %.2f edge_bwd %.2f' % (k['edge_message_fwd'], k['edge_message_bwd']))" | tee -a $O/ab_bwd_wide.txt
done; done; done
