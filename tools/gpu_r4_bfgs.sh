#!/bin/bash
O=gpurun_out/r04_bfgs; mkdir -p $O
timeout 1500 python -m pytest tests/test_bfgs.py tests/test_relax.py tests/test_cg.py tests/test_host_opt.py tests/test_mc_gpu.py tests/test_eam.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -8
python tools/bench_relax.py > $O/bench_relax.jsonl 2> $O/err.txt
python - $O/bench_relax.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['optimizer'], d['fmax'], d['converged'], d['chain_evaluations_needed'], d['wall_s'], d['ms_per_256_chain_evaluations'], d['full_batch_evaluation_ms'])
PY
python tools/bench_mc.py --chains 256 --relax-steps 20 --steps 10 > $O/bench_mc.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/bench_mc.json')); print(d['s_per_lockstep'], d['proposals_per_s'], d['split_s_per_lockstep'])"
