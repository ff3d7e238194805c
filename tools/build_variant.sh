#!/bin/bash
# Build a variant of libvssr_eval.so for same-box A/B runs (tools/gpu_ab.sh): tools/build_variant.sh <name> [XFLAGS...]
# Sources are copied to build/variants/src_<name> so the product objects are left alone; output build/variants/lib_<name>.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build/variants/src_$name
mkdir -p $d/surface-sampling_amd/csrc $d/include
cp $root/surface-sampling_amd/csrc/*.hip $root/surface-sampling_amd/csrc/*.h $root/surface-sampling_amd/csrc/Makefile $d/surface-sampling_amd/csrc/
cp $root/include/*.h $d/include/
make -C $d/surface-sampling_amd/csrc -j8 -s XFLAGS="$*" OUT=$root/build/variants/lib_$name.so
ls -la $root/build/variants/lib_$name.so
