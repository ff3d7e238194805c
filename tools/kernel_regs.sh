#!/bin/bash
# Register / spill / LDS summary of every kernel of one csrc file: tools/kernel_regs.sh painn_node_mfma.hip [XFLAGS...]
f=$1; shift
cd "$(dirname "$0")/../surface-sampling_amd/csrc"
extra=""; [ "$f" = painn_edge_mfma.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $extra "$@" -S --cuda-device-only -o /tmp/kregs.s $f 2>/dev/null
python3 - <<'PY'
import re
txt = open('/tmp/kregs.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    pass
# amdhsa metadata block
meta = txt[txt.rfind('amdhsa.kernels'):]
for blk in meta.split('  - .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk)
    import subprocess
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    print(f"{dem[:60]:60s} vgpr {g('vgpr_count').group(1):>4s} spill {g('vgpr_spill_count').group(1):>3s} sgpr {g('sgpr_count').group(1):>4s} sspill {g('sgpr_spill_count').group(1):>3s} lds {g('group_segment_fixed_size').group(1):>6s} scratch {g('private_segment_fixed_size').group(1)}")
PY
