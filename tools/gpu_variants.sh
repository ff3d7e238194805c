#!/bin/bash
# debug helper: run a test subset against several builds of the library (gpu_variants/lib_*.so)
for f in gpu_variants/lib_*.so; do
  cp $f surface-sampling_amd/libvssr_eval.so
  echo "== $f"
  timeout 600 python -m pytest tests -m gpu -q -k "${KEXPR:-determinism or kat or batched}" 2>&1 | tail -6
done
