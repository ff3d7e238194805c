#!/bin/bash
# rocprofv3 kernel trace of the bench (same command as the bench line), summary into gpurun_out/prof_*.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2> gpurun_out/prof_bench.err
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_kernel_stats.csv && head -30 "$f"
# keep only the small summaries (traces can be large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
cat gpurun_out/prof_bench.log | cut -c1-600
tail -3 gpurun_out/prof_bench.err
