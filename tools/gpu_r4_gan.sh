#!/bin/bash
O=gpurun_out/r04_gan; mkdir -p $O
timeout 900 python -m pytest tests/test_cg.py tests/test_eam.py tests/test_gpu_parity.py tests/test_gpu_abi.py -m gpu -x -q -k "gan or cg or eam or tersoff or lammps" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
VSSR_TERSOFF_SITE=1 timeout 900 python -m pytest tests/test_cg.py tests/test_gpu_parity.py -m gpu -x -q -k "gan or cg or tersoff or lammps" > $O/pytest_one_thread.log 2>&1; tail -2 $O/pytest_one_thread.log
timeout 1200 python tools/bench_gan.py --chains 256,1024,4096 --steps 4 > $O/bench_gan.jsonl 2> $O/err; echo rc=$?
VSSR_TERSOFF_SITE=1 timeout 1200 python tools/bench_gan.py --chains 256,4096 --steps 4 > $O/bench_gan_one_thread.jsonl 2>> $O/err; echo rc=$?
cat $O/bench_gan.jsonl $O/bench_gan_one_thread.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['chains'], round(d['proposals_per_s'], 1), round(d['s_per_lockstep'], 3), round(d['acceptance'], 3), round(d['mean_energy_eV'], 6), round(d['speedup_vs_reference_per_proposal'], 1))"
tail -5 $O/err
