#!/bin/bash
O=gpurun_out/ab_streams; mkdir -p $O
for rep in 1 2; do for s in 2 3 4 1; do
python bench.py --steps 60 --warmup 10 --streams $s --profile-steps 2 > $O/s${s}_$rep.json 2>> $O/err
python3 -c "import json; d=json.load(open('$O/s${s}_$rep.json')); print($s, round(d['value'],1), round(d['ms_per_step'],3))"
done; done
