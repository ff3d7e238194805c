#!/bin/bash
# the neighbor kernels with 64 / 32 lanes per centre forced over the whole GPU suite (the default, 16, is the plain run), then the benches
O=gpurun_out/nbr_lpc; mkdir -p $O
for l in 64 32; do
VSSR_NBR_LPC=$l timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_lpc$l.log 2>&1; echo lpc $l: $(grep -h "passed\|failed" $O/pytest_lpc$l.log)
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo default: $(grep -h "passed\|failed" $O/pytest.log)
timeout 600 python tools/bench_gan.py --chains 256,1024,4096 --steps 4 2>/dev/null > $O/bench_gan.jsonl; python3 -c "
import sys, json
for l in open('$O/bench_gan.jsonl'):
    d = json.loads(l); print('gan', d['chains'], round(d['proposals_per_s'], 1), round(d['s_per_lockstep'], 4))"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', round(d['value'], 1), round(d['ms_per_step'], 3), d['kernel_ms_per_step']['neighbor_list'])"
