#!/bin/bash
# round 5: live-chain compaction of the CG minimiser -- tests, then the GaN loop with and without it on the same box
O=gpurun_out/r5_c; mkdir -p $O
timeout 1200 python -m pytest tests/test_cg.py tests/test_relax.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for flag in 0 1 0 1; do
  VSSR_RELAX_COMPACT=$flag timeout 900 python tools/bench_gan.py --chains 256,1024,4096 --steps 4 2>>$O/bench_gan.err | tee -a $O/bench_gan_compact$flag.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); w = d['lockstep_waste']; print('gan compact=$flag', d['chains'], round(d['proposals_per_s'], 1), 'dispatched/needed', round(w['dispatched_over_needed'], 3), 'acc', round(d['acceptance'], 4), 'E', d['mean_energy_eV'])"
done
tail -3 $O/bench_gan.err
