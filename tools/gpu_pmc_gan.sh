#!/bin/bash
# PMC passes over the Tersoff / neighbor kernels of the GaN workload (tools/bench_gan.py, 4096 chains, one MC step)
O=gpurun_out/pmc_gan
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run_pass () {
  name=$1; shift
  rm -rf $O/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "k_tersoff_site|k_nbr|k_scan_rows|k_rev" --output-format csv -d $O/pmc_$name -o p -- python3 tools/bench_gan.py --chains 4096 --steps 1 > /dev/null 2> $O/pmc_$name.err
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT
python3 - <<'PY'
import csv, glob, collections
for name in ("sq1", "sq2"):
    f = glob.glob(f"gpurun_out/pmc_gan/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not f: print(name, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
tail -2 $O/*.err
