#!/bin/bash
# quick regression after a change to shared kernels: the whole GPU suite, the GaN bench, a short headline run
O=gpurun_out/quick; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/bench_gan.py --chains 256,4096 --steps 4 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('gan', d['chains'], round(d['proposals_per_s'], 1), round(d['s_per_lockstep'], 4))"
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', round(d['value'], 1), round(d['ms_per_step'], 3), d['kernel_ms_per_step'])"
