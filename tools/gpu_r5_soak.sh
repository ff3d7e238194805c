#!/bin/bash
# round 5: repeatability soak of the final build (every neighbor-sum class incl. the multi-pass forms, the benchmark batch, long run)
O=gpurun_out/r5_soak; mkdir -p $O
NCHAIN=96 REPS=60 python tools/gpu_stress_classes.py 2>/dev/null | tee $O/soak_classes.txt
REPS=400 python tools/gpu_stress_bench.py 2>/dev/null | tee $O/soak_bench.txt
python tools/gpu_longrun.py 2>/dev/null | tail -3 | tee $O/longrun.txt
