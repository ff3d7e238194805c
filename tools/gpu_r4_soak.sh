#!/bin/bash
# Round-4 repeatability soak of the final build: every neighbor-sum class incl. the 4-feature slices + the benchmark batch + longrun
O=gpurun_out/r04_soak; mkdir -p $O
NCHAIN=96 REPS=60 timeout 2400 python tools/gpu_stress_classes.py > $O/soak_classes.txt 2>&1; tail -9 $O/soak_classes.txt
REPS=400 timeout 1200 python tools/gpu_stress_bench.py > $O/soak_bench.txt 2>&1; tail -2 $O/soak_bench.txt
timeout 1200 python tools/gpu_longrun.py > $O/longrun.txt 2>&1; tail -4 $O/longrun.txt
