#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, a short bench; logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep -m1 "Model name" >> gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
