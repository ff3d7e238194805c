#!/bin/bash
# round-4 baseline: MC loop split, relaxation driver cost, default bench line on the r3 build
set -x
O=gpurun_out/r04_base; mkdir -p $O
python tools/bench_mc.py --chains 256 --relax-steps 20 --steps 10 > $O/bench_mc.json 2> $O/bench_mc.err
python tools/bench_relax.py > $O/bench_relax.jsonl 2> $O/bench_relax.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 2 > $O/bench_n1_s2.json 2> $O/bench_n1_s2.err
tail -3 $O/*.json $O/*.jsonl; tail -5 $O/*.err
