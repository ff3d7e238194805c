#!/bin/bash
# mc.ConcurrentChains: MC proposals/s with the chains in 1 / 2 / 3 concurrent groups, single-point and relaxed acceptance energies
O=gpurun_out/r04_groups; mkdir -p $O
timeout 600 python -m pytest tests/test_mc_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
python tools/bench_mc.py --chains 256 --steps 20 --no-relax > $O/norelax_g1_$rep.json 2>> $O/err
python tools/bench_mc.py --chains 256 --steps 20 --no-relax --groups 2 > $O/norelax_g2_$rep.json 2>> $O/err
python tools/bench_mc.py --chains 256 --steps 20 --no-relax --groups 3 > $O/norelax_g3_$rep.json 2>> $O/err
done
python tools/bench_mc.py --chains 256 --steps 6 --relax-steps 20 > $O/relax_g1.json 2>> $O/err
python tools/bench_mc.py --chains 256 --steps 6 --relax-steps 20 --groups 2 > $O/relax_g2.json 2>> $O/err
for f in $O/*.json; do echo $f; python3 -c "import json,sys; d=json.load(open('$f')); print(d.get('groups',1), round(d['proposals_per_s'],1), round(d['s_per_lockstep']*1e3,2))"; done
tail -3 $O/err
