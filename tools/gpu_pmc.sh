#!/bin/bash
# PMC counter passes (own runs, no tracing flags) for kernels matching $1; summaries -> gpurun_out/pmc_*.txt
KREGEX=${1:-k_edge_fwd_mfma}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run_pass () {
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d gpurun_out/pmc_$name -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_$name.err
}
run_pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY
run_pass c SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ
python3 - <<'PY'
import csv, glob, collections
for name in "abc":
    files = glob.glob(f"gpurun_out/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(name, k)
        for c, v in d.items():
            print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
