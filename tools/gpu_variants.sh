#!/bin/bash
# debug helper: run a command against several builds of the library (gpu_variants/lib_*.so)
cp surface-sampling_amd/libvssr_eval.so /tmp/lib_keep.so
for f in gpu_variants/lib_*.so; do
  cp $f surface-sampling_amd/libvssr_eval.so
  echo "== $f"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  REPS=${REPS:-15} timeout 600 python tools/gpu_stress.py 2>&1 | grep evaluations
done
cp /tmp/lib_keep.so surface-sampling_amd/libvssr_eval.so
