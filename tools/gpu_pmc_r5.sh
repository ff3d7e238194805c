#!/bin/bash
# Round-5 PMC collection for bench.py's `roofline.binding_resource`: every pass is its own rocprofv3 run with --pmc only
# (no tracing flags), one bench step after one warm-up.  Writes gpurun_out/r05_pmc/pmc_summary.json (raw per-launch means
# + derived percentages per kernel); copy it to profiles/r05/pmc_summary.json.
#   usage: bash tools/gpu_pmc_r5.sh [kernel regex]
KREGEX=${1:-'k_edge_(fwd|bwd)_mfma|k_update_(fwd|bwd)_mfma|k_reduce_gpart'}
O=gpurun_out/r05_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run_pass () {
  name=$1; shift
  rm -rf $O/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d $O/pmc_$name -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2> $O/pmc_$name.err
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run_pass ta TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
python3 tools/pmc_summarize.py $O > $O/pmc_summary.txt
cat $O/pmc_summary.txt
tail -2 $O/*.err
