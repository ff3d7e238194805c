#!/bin/bash
# round 4, first build: gpu tests + the new bench line (2 streams / 1 stream) + PMC summary
O=gpurun_out/r04_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s2.json 2> $O/bench_s2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 3 > $O/bench_s3.json 2> $O/bench_s3.err
bash tools/gpu_pmc_r4.sh > $O/pmc.log 2>&1
cp gpurun_out/r04_pmc/pmc_summary.json gpurun_out/r04_pmc/pmc_summary.txt $O/ 2>/dev/null
tail -n 15 $O/pytest_gpu.log
