#!/bin/bash
# debug helper: smoke + repeatability stress against several builds of the library (build/variants/lib_*.so)
for f in build/variants/lib_*.so; do
  echo "== $f"
  VSSR_EVAL_LIB=$PWD/$f timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  VSSR_EVAL_LIB=$PWD/$f REPS=${REPS:-15} timeout 600 python tools/gpu_stress.py 2>&1 | grep evaluations
done
