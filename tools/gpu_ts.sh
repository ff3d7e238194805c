#!/bin/bash
# Tersoff site tile with batched LDS reads: tests, phase clocks, same-box A/B of the GaN loop (build/variants/lib_ts_old.so / lib_ts_new.so)
O=gpurun_out/ts; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "tersoff or cg or lammps or gan or Tersoff or chain_resident or compaction" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
VSSR_EVAL_LIB=$PWD/build/variants/lib_cmphase.so timeout 300 python tools/gpu_cm_phase.py 1 2>&1 | grep -v amdgpu.ids | tee $O/cm_phase.txt
for rep in 1 2; do
for v in ts_old ts_new; do
  VSSR_EVAL_LIB=$PWD/build/variants/lib_$v.so timeout 900 python tools/bench_gan.py --chains 1,256,1024,4096,16384 --steps 4 2>>$O/bench_gan.err | tee -a $O/bench_gan_$v.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('gan $v', d['chains'], round(d['proposals_per_s'], 1), 'acc', round(d['acceptance'], 4), 'E', d['mean_energy_eV'])"
done; done
tail -3 $O/bench_gan.err
