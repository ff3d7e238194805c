#!/bin/bash
# the whole GPU suite with its log kept (gpurun_out/suite/pytest_gpu.log) -> profiles/rNN/pytest_gpu.log
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
O=gpurun_out/suite; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=6 "$@" > $O/pytest_gpu.log 2>&1; rc=$?
grep -E "passed|failed|error|^FAILED|^ERROR|s call" $O/pytest_gpu.log | tail -20
grep -A3 "deviation from the fp64 oracle" $O/pytest_gpu.log
grep "fmax - print" $O/pytest_gpu.log
exit $rc
