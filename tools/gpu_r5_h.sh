#!/bin/bash
# round 5: multi-pass forward form -- waves per workgroup A/B on large chains, then tests with the product build
O=gpurun_out/r5_h; mkdir -p $O
for rep in 1 2; do for f in build/variants/lib_w08.so build/variants/lib_w12.so; do for atoms in 1000 1400; do
  VSSR_EVAL_LIB=$PWD/$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --chains-per-gpu 128 --atoms-per-chain $atoms 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('$f atoms/chain $atoms', 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/ab_sub16_waves.txt
done; done; done
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "narrow or mixed or large_chain or repeatability or fallback" > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
