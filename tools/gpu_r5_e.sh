#!/bin/bash
# round 5: the chain-resident CG minimiser -- tests, then the GaN loop with the lock-step driver and with it on the same box
O=gpurun_out/r5_e; mkdir -p $O
timeout 1500 python -m pytest tests/test_cg.py tests/test_relax.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cg or tersoff or gan or lammps or relax" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for flag in 0 2 1 0 2 1; do
  VSSR_CG_FUSED=$flag timeout 900 python tools/bench_gan.py --chains 1,256,1024,4096,16384 --steps 3 2>>$O/bench_gan.err | tee -a $O/bench_gan_fused$flag.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); w = d['lockstep_waste']; print('gan fused=$flag', d['chains'], round(d['proposals_per_s'], 1), 's/step', round(d['s_per_lockstep'], 5), 'dispatched/needed', round(w['dispatched_over_needed'], 3), 'acc', round(d['acceptance'], 4), 'E', d['mean_energy_eV'])"
done
tail -3 $O/bench_gan.err
