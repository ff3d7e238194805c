#!/bin/bash
# round 6: phase clocks of the edge kernels under both K layouts (build/ab_ph/lib_ph{0,1}.so: -DEDGE_PHASE_TIMING with
# -DEDGE_KDENSE=0/1), then removal ablations on the K-dense kernels (build/variants/lib_a_*.so), one engine per GPU
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
O=gpurun_out/r6b; mkdir -p $O
for k in 0 1; do
  echo "== EDGE_KDENSE=$k" >> $O/edge_phase_clocks.txt
  VSSR_EVAL_LIB=$PWD/build/ab_ph/lib_ph$k.so python tools/gpu_edge_phase.py >> $O/edge_phase_clocks.txt 2>$O/phase$k.err
done
cat $O/edge_phase_clocks.txt
rm -f gpurun_out/ab1.log
AB_REPS=2 AB_STEPS=10 bash tools/gpu_ab1.sh
cp gpurun_out/ab1.log $O/ablations_kdense.txt
