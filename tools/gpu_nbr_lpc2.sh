#!/bin/bash
# neighbor kernels: lanes per centre (k_nbr / k_rev separately) on the PaiNN bench at three chain sizes
for a in 80 260; do for cfg in "64 64" "16 16" "16 64" "32 32" "32 16" "16 32" "64 16"; do
set -- $cfg
c=256; [ $a = 80 ] && c=1024
VSSR_NBR_LPC=$1 VSSR_REV_LPC=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --atoms-per-chain $a --chains-per-gpu $c --streams 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('atoms $a nbr $1 rev $2', round(d['value'], 1), round(d['ms_per_step'], 3), 'nbr', round(d['kernel_ms_per_step']['neighbor_list'], 4))"
done; done
