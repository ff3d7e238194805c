#!/bin/bash
# 4-feature-slice class: parity tests, repeatability soak of the new instantiations, per-atom cost at 700 / 1000 / 1400 atoms
O=gpurun_out/r04_fs4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrow or mixed or fallback or batching or repeatab" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
ONLY=fs4all,big4,fs8all NCHAIN=24 REPS=10 timeout 900 python tools/gpu_stress_classes.py > $O/soak.txt 2>&1; tail -5 $O/soak.txt
for n in 260 700 1000 1400; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 --atoms-per-chain $n --chains-per-gpu ${CHAINS:-128} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('atoms/chain $n', 'atoms', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes.txt
done
