#!/bin/bash
O=gpurun_out/r04_mc; mkdir -p $O
timeout 900 python -m pytest tests/test_mc_gpu.py tests/test_gpu_wrappers.py tests/test_eam.py tests/test_cg.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/bench_mc.py --chains 256 --relax-steps 20 --steps 10 > $O/bench_mc.json 2> $O/bench_mc.err
python tools/bench_mc.py --chains 256 --steps 20 --no-relax > $O/bench_mc_norelax.json 2>> $O/bench_mc.err
tail -c 900 $O/bench_mc.json; echo; tail -c 900 $O/bench_mc_norelax.json; tail -3 $O/bench_mc.err
