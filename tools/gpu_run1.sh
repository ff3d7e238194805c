#!/bin/bash
# GPU iteration: parity tests on the product library, then same-box A/B of build/variants
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
rm -f gpurun_out/ab.log
AB_STEPS=${AB_STEPS:-10} AB_REPS=${AB_REPS:-2} bash tools/gpu_ab.sh
