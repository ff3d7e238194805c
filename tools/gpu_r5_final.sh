#!/bin/bash
# round 5: the measurements that go to profiles/r05 -- whole GPU suite, headline bench (two / one engine), rocprofv3 kernel stats of both,
# PMC passes, the energy-word table
O=gpurun_out/r5_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python3 -c "
import json; d = json.load(open('$O/bench_n1.json')); print('bench', round(d['value'], 1), round(d['ms_per_step'], 3), d['kernel_ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['executed_pipe']['frac'], d['cpu_baseline']['value'])"
python bench.py --streams 1 --no-cpu-baseline > $O/bench_n1_streams1.json 2>> $O/bench_n1.err
for s in 1 2; do
  rm -rf $O/prof_s$s
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s$s -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams $s > $O/bench_under_rocprof_streams$s.json 2> $O/rocprof_s$s.err
  f=$(find $O/prof_s$s -name '*kernel_stats.csv' | head -1); cp "$f" $O/rocprof_kernel_stats_streams$s.csv; head -8 $O/rocprof_kernel_stats_streams$s.csv
  rm -rf $O/prof_s$s
done
bash tools/gpu_pmc_r5.sh > $O/pmc.log 2>&1; tail -12 $O/pmc.log
python tools/gpu_energy_words.py > $O/energy_words.jsonl 2>/dev/null; cat $O/energy_words.jsonl
python tools/bench_mc.py > $O/bench_mc.json 2>/dev/null; cat $O/bench_mc.json | cut -c1-600
