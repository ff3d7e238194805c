#!/bin/bash
# Round-4 evidence run (everything under gpurun_out/r04_final; copy what should be judged into profiles/r04/).
#   1 GPU tests   2 default bench line (2 engines per GPU) + the single-stream line   3 rocprofv3 kernel stats of both commands
#   4 PMC passes -> pmc_summary.json   5 MC loop split + relaxation driver cost   6 secondary lines (chain sizes)   7 full parity
O=gpurun_out/r04_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 900 python bench.py --streams 1 --no-cpu-baseline > $O/bench_n1_streams1.json 2> $O/bench_n1_streams1.err
python3 - $O <<'PY'
import json, sys
for f in ("bench_n1.json", "bench_n1_streams1.json"):
    d = json.loads(open(sys.argv[1] + "/" + f).read().strip().splitlines()[-1])
    print(f, 'value %.0f  ms/step %.3f  streams %d  roofline.frac %.3f  single-stream ms %.3f' % (d['value'], d['ms_per_step'], d['config']['streams_per_gpu'], d['roofline']['frac'], d['single_stream']['ms_per_step']))
    print('   ', {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 2 1; do
  rm -rf $O/prof_s$s
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s$s -o bench -- python3 bench.py --no-cpu-baseline --streams $s > $O/bench_under_rocprof_streams$s.json 2> $O/prof_s$s.err
  f=$(find $O/prof_s$s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_kernel_stats_streams$s.csv
  rm -rf $O/prof_s$s
done
head -8 $O/rocprof_kernel_stats_streams1.csv | cut -c1-160
bash tools/gpu_pmc_r4.sh > $O/pmc.log 2>&1
cp gpurun_out/r04_pmc/pmc_summary.json gpurun_out/r04_pmc/pmc_summary.txt $O/ 2>/dev/null
python tools/bench_mc.py --chains 256 --relax-steps 20 --steps 10 > $O/bench_mc.json 2> $O/bench_mc.err
python tools/bench_mc.py --chains 256 --relax-steps 20 --steps 10 --no-relax > $O/bench_mc_norelax.json 2>> $O/bench_mc.err
python tools/bench_relax.py > $O/bench_relax.jsonl 2> $O/bench_relax.err
for n in 260 380 480 700 1000; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --atoms-per-chain $n --chains-per-gpu 256 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']; a = d['config']['atoms_per_gpu']
print('atoms/chain $n', 'atoms', a, 'evals/s %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'us/atom %.4f' % (1e3 * d['ms_per_step'] / a), ' '.join('%s=%.2f' % (n[:12], v) for n, v in k.items()))" | tee -a $O/bench_chain_sizes.txt
done
NCHAIN=256 python tools/gpu_full_parity.py > $O/full_parity_256chains.json 2> $O/full_parity.err
tail -2 $O/bench_mc.json $O/full_parity_256chains.json | cut -c1-600
