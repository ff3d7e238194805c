#!/bin/bash
# PMC passes for the edge kernels (current build) + the list of available counters
KREGEX=${1:-k_edge_bwd_mfma}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
[ -f gpurun_out/avail.txt ] || rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1
run_pass () {
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d gpurun_out/pmc_$name -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_$name.err
}
run_pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY
run_pass c SQ_WAVES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC
run_pass d TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_TA_TCP_STATE_READ_sum TA_TA_BUSY_sum
python3 - <<'PY'
import csv, glob, collections
for name in "abcd":
    files = glob.glob(f"gpurun_out/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(name, k)
        for c, v in d.items():
            print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
tail -3 gpurun_out/pmc_d.err
