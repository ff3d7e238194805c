#!/bin/bash
for a in 0 1 2 4 7 8 16 24 31; do
  VSSR_ABLATE=$a timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('ablate=$a update_fwd %.3f ms/step (3 launches)' % (k['update_fwd']))"
done
