#!/bin/bash
# round 5: GaN loop, final build -- tests, default driver choice vs forced lock-step / forced chain-resident, phase clocks
O=gpurun_out/r5_f; mkdir -p $O
timeout 1500 python -m pytest tests/test_cg.py tests/test_relax.py tests/test_gpu_parity.py tests/test_eam.py tests/test_mc_gpu.py -m gpu -q -p no:cacheprovider -k "cg or tersoff or gan or lammps or relax or eam or Cu or mc" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for rep in 1 2; do
for flag in default 0 1; do
  if [ $flag = default ]; then unset VSSR_CG_FUSED; else export VSSR_CG_FUSED=$flag; fi
  timeout 900 python tools/bench_gan.py --chains 1,256,1024,4096,16384 --steps 4 2>>$O/bench_gan.err | tee -a $O/bench_gan_$flag.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); w = d['lockstep_waste']; print('gan driver=$flag', d['chains'], round(d['proposals_per_s'], 1), 's/step', round(d['s_per_lockstep'], 5), 'dispatched/needed', round(w['dispatched_over_needed'], 3), 'acc', round(d['acceptance'], 4), 'E', d['mean_energy_eV'])"
done; done
unset VSSR_CG_FUSED
VSSR_EVAL_LIB=$PWD/build/variants/lib_cmphase.so python tools/gpu_cm_phase.py 1 > $O/cm_phase.txt 2>/dev/null; VSSR_EVAL_LIB=$PWD/build/variants/lib_cmphase.so python tools/gpu_cm_phase.py 512 >> $O/cm_phase.txt 2>/dev/null; cat $O/cm_phase.txt
