#!/bin/bash
# round 5: compaction (threshold form) test + GaN at 4096 / 16384 chains with / without it; then all 2 048 chains of configs[4] vs the oracle
O=gpurun_out/r5_d; mkdir -p $O
timeout 600 python -m pytest tests/test_cg.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for flag in 0 1 0 1; do
  VSSR_RELAX_COMPACT=$flag timeout 900 python tools/bench_gan.py --chains 256,4096,16384 --steps 3 2>>$O/bench_gan.err | tee -a $O/bench_gan_compact$flag.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); w = d['lockstep_waste']; print('gan compact=$flag', d['chains'], round(d['proposals_per_s'], 1), 'dispatched/needed', round(w['dispatched_over_needed'], 3), 'acc', round(d['acceptance'], 4), 'E', d['mean_energy_eV'])"
done
NBLOCKS=8 timeout 2400 python tools/gpu_full_parity.py > $O/full_parity_2048chains.json 2> $O/full_parity.err; tail -c 1500 $O/full_parity_2048chains.json; tail -3 $O/full_parity.err
