#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/fullparity
( time NBLOCKS=8 python tools/gpu_full_parity.py ) > gpurun_out/fullparity/full_parity_2048chains.json 2> gpurun_out/fullparity/log.txt
tail -3 gpurun_out/fullparity/log.txt; cut -c1-700 gpurun_out/fullparity/full_parity_2048chains.json
