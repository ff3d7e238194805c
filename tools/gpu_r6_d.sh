#!/bin/bash
# round 6: GPU suite on the current build, then same-box A/B of build/variants/lib_*.so (one engine per GPU)
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
bash tools/gpu_suite.sh
rm -f gpurun_out/ab1.log
AB_REPS=${AB_REPS:-3} AB_STEPS=${AB_STEPS:-20} bash tools/gpu_ab1.sh
mkdir -p gpurun_out/r6d; cp gpurun_out/ab1.log gpurun_out/r6d/ab.txt
