#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_gan; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o gan -- python3 tools/bench_gan.py --chains 4096 --steps 3 > $O/line.json 2> $O/err
f=$(find $O -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-200
cat $O/line.json | cut -c1-400
