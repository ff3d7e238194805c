#!/bin/bash
# A/B of runtime knobs (env) on the same box: KNOB="VSSR_L0_FACTORISE" VALUES="1 0"
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
for v in $VALUES; do
  echo "== $KNOB=$v parity"; env $KNOB=$v timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
done
for rep in 1 2; do for v in $VALUES; do
  env $KNOB=$v python bench.py --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$KNOB=$v', 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a gpurun_out/ab.log
done; done
