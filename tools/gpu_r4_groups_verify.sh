#!/bin/bash
# soak of the threaded path: 256 chains in 3 groups, 40 single-point MC steps and 4 relaxed ones, then the same steps with one ensemble
O=gpurun_out/r04_groups; mkdir -p $O
python tools/bench_mc.py --chains 256 --steps 40 --no-relax --groups 3 --verify > $O/verify_norelax.json 2> $O/verify.err; echo rc=$?
python tools/bench_mc.py --chains 256 --steps 4 --relax-steps 20 --groups 3 --verify > $O/verify_relax.json 2>> $O/verify.err; echo rc=$?
python tools/bench_mc.py --chains 96 --steps 30 --no-relax --groups 4 --verify > $O/verify_norelax_96x4.json 2>> $O/verify.err; echo rc=$?
cat $O/verify_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['chains'], d['groups'], d['relax_steps'], round(d['proposals_per_s'], 1), d.get('identical_to_one_ensemble'))"
tail -3 $O/verify.err
