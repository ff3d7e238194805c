#!/bin/bash
# round 5, second call: the fixed tests, the 16-atom update_bwd A/B, lock-step waste of the GaN CG loop and of PaiNN relaxations
O=gpurun_out/r5_b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wrappers.py tests/test_relax.py "tests/test_sharding.py::test_device_results_reach_torch_and_rccl_without_a_host_copy" -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
rm -f gpurun_out/ab1.log
AB_REPS=2 AB_STEPS=20 bash tools/gpu_ab1.sh > $O/ab_update_bwd_rt1.txt 2>&1; cat $O/ab_update_bwd_rt1.txt
timeout 900 python tools/bench_gan.py --chains 256,4096 --steps 4 > $O/bench_gan.jsonl 2>$O/bench_gan.err; cat $O/bench_gan.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('gan', d['chains'], round(d['proposals_per_s'], 1), json.dumps(d['lockstep_waste']))"
tail -3 $O/bench_gan.err
timeout 900 python tools/bench_relax.py > $O/bench_relax.jsonl 2>$O/bench_relax.err; cat $O/bench_relax.jsonl
