#!/bin/bash
# round 6: whole GPU suite (new: all 256 bench chains against the committed fp64 vectors, position_dtype, per-rank diagnostics),
# then a same-box ablation: three table loads per reverse step instead of four (build/variants/lib_a_3loads.so, results wrong)
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
O=gpurun_out/r6c; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
grep -A3 "deviation from the fp64 oracle" $O/pytest_gpu.log
grep "fmax - print" $O/pytest_gpu.log
rm -f gpurun_out/ab1.log
AB_REPS=3 AB_STEPS=10 bash tools/gpu_ab1.sh
cp gpurun_out/ab1.log $O/ablation_3loads.txt
