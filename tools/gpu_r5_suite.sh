#!/bin/bash
# round 5: the whole GPU suite (keeps going after a failure), then a short headline run
O=gpurun_out/r5_suite; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -25 $O/pytest.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>$O/bench.err | tee $O/bench.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', round(d['value'], 1), round(d['ms_per_step'], 3), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['executed_pipe']['frac'])"
tail -5 $O/bench.err
