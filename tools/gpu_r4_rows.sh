#!/bin/bash
# Row-scaling experiment: the stand-alone message MLP (non-factorised layer 0: VSSR_L0_FACTORISE=0 launches it once per step)
# with 32- and 64-atom tiles, rolled and pipelined GEMMs, at its own occupancy (2 workgroups per CU) and with ONE workgroup
# per CU (the occupancy of the fused update kernels); prints the message_mlp class time per step.
O=gpurun_out/r04_rows; mkdir -p $O
for one in 0 1; do for rt in 2 4; do for pf in 0 1; do
  VSSR_L0_FACTORISE=0 VSSR_MSG_MLP_ONE_WG=$one VSSR_MSG_MLP_RT=$rt VSSR_MSG_MLP_PF=$pf python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 > $O/b_${one}_${rt}_${pf}.json 2> $O/b_${one}_${rt}_${pf}.err
  python - $O/b_${one}_${rt}_${pf}.json $one $rt $pf <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
k = d["kernel_ms_per_step"]
print(f"one_wg_per_cu={sys.argv[2]} rows={16 * int(sys.argv[3])} pipelined={sys.argv[4]}  message_mlp {k.get('message_mlp'):.4f} ms/step   (update_fwd {k.get('update_fwd'):.4f}, step {d['device_ms_per_step']:.3f})")
PY
done; done; done | tee $O/summary.txt
