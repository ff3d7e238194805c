#!/bin/bash
# PMC collection for bench.py's `roofline.from_committed_profile`: every pass is its own rocprofv3 run with --pmc only (no tracing
# flags), one bench step after one warm-up.  Writes gpurun_out/pmc/pmc_summary.{json,txt} (raw per-launch means + derived
# percentages per kernel + the digest of the kernel sources); copy both to profiles/rNN/.
#   usage: bash tools/gpu_pmc.sh [kernel regex]
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root" || exit 1
KREGEX=${1:-'k_edge_(fwd|bwd)_mfma|k_update_(fwd|bwd)_mfma|k_reduce_gpart'}
O="$root/gpurun_out/pmc"
mkdir -p "$O"
export TMPDIR=/tmp
rc=0
run_pass () {
  name=$1; shift
  rm -rf "$O/pmc_$name"
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d "$O/pmc_$name" -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2> "$O/pmc_$name.err"
  if ! find "$O/pmc_$name" -name '*counter_collection.csv' | grep -q .; then
    echo "gpu_pmc.sh: pass $name produced no counter CSV:" >&2; tail -3 "$O/pmc_$name.err" >&2; rc=1
  fi
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run_pass ta TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
[ $rc -eq 0 ] || exit $rc
python3 tools/pmc_summarize.py "$O" > "$O/pmc_summary.txt" || exit 1
cat "$O/pmc_summary.txt"
