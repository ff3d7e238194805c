#!/bin/bash
# round 6, first contact: whole GPU suite on the K-dense build (product library), then a same-box A/B of the two K layouts
# (build/variants/lib_kd0.so = EDGE_KDENSE=0, lib_kd1.so = EDGE_KDENSE=1; tools/build_variant.sh), one engine per GPU
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
O=gpurun_out/r6a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
rm -f gpurun_out/ab1.log
AB_REPS=3 AB_STEPS=20 bash tools/gpu_ab1.sh
cp gpurun_out/ab1.log $O/ab_kdense.txt
