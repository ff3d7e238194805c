#!/bin/bash
# one PMC pass over ALL kernels: how busy are the L1 address / tag units (TA, TCP)?
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_ta
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_ta -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_ta.err
python3 - <<'PY'
import csv, glob, collections
files = glob.glob("gpurun_out/pmc_ta/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(files[0])):
    acc[row["Kernel_Name"].split("(")[0][:44]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"{'kernel':44s} {'n':>3s} {'cyc/launch':>10s} {'TA busy%':>8s} {'acc/cyc/CU':>10s} {'VALU/cyc/SIMD':>12s} {'acc/vmem':>8s}")
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1]["GRBM_GUI_ACTIVE"])):
    n = len(d["GRBM_GUI_ACTIVE"])
    cyc = sum(d["GRBM_GUI_ACTIVE"]) / n / 8.0
    ta = sum(d["TA_TA_BUSY_sum"]) / n / 256.0 / cyc * 100
    a = sum(d["TCP_TOTAL_CACHE_ACCESSES_sum"]) / n / 256.0 / cyc
    v = sum(d["SQ_INSTS_VALU"]) / n / 1024.0 / cyc * 4
    vm = (sum(d["SQ_INSTS_VMEM_RD"]) + sum(d["SQ_INSTS_VMEM_WR"])) / n
    print(f"{k:44s} {n:3d} {cyc:10.0f} {ta:8.1f} {a:10.2f} {v:12.2f} {sum(d['TCP_TOTAL_CACHE_ACCESSES_sum'])/n/max(vm,1):8.1f}")
PY
tail -2 gpurun_out/pmc_ta.err
