#!/bin/bash
# round 5: two-pass forward neighbor sum for 788 .. 1 462-atom chains -- parity tests, then chain-size lines with / without it
O=gpurun_out/r5_g; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "narrow or mixed or large_chain or repeatability or fallback" > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
for flag in 1 0 1 0; do
for atoms in 700 1000 1400; do
  VSSR_EDGE_BWD_MPASS=$flag VSSR_EDGE_FWD_2PASS=$((16*flag)) python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --chains-per-gpu 128 --atoms-per-chain $atoms 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('multipass=$flag atoms/chain $atoms atoms', a, 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes_2pass.txt
done; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --chains-per-gpu 128 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('atoms/chain 260 atoms', a, 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes_2pass.txt
