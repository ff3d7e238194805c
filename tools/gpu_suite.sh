#!/bin/bash
# the whole GPU suite with its log kept: gpurun_out/suite/pytest_gpu.log (summary line + slowest tests printed)
O=gpurun_out/suite; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 "$@" > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error|^FAILED|^ERROR|s call" $O/pytest_gpu.log | tail -20
