#!/bin/bash
# Kernel trace + counter passes for ONE render variant of bench.py (run on the GPU box from the repo root):
#   bash tools/collect_pmc_variant.sh <tag> <variant> [quick]     e.g.  r03_tcnn_f16 tcnn_f16
# -> gpurun_out/pmc_<tag>/<pass>/ ; tools/summarise_pmc.py <tag> pmc_variant folds them into profiles/ (run it where
# profiles/ is tracked: gpurun only merges gpurun_out/ back).  One rocprofv3 run per counter group (PMC passes serialise
# kernels; never combined with API traces); at most two TA and four TCP counters per pass (more: "exceeds the capabilities
# of the hardware", and the aborted profiler then hangs), every pass under its own timeout.
TAG=${1:-r03_tcnn_f16}
VAR=${2:-tcnn_f16}
QUICK=${3:-}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
B="$ROOT/bench.py --variant $VAR --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <dir> <counters...>
  local d=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$d -- python3 $B > $O/$d.log 2>&1 || echo "pass $d failed" >> $O/failed.log
  echo "pass $d done" >> $O/progress.log
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ROOT/bench.py --variant $VAR --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr
pass ta TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
if [ -z "$QUICK" ]; then
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_max
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TD_TCP_STALL_CYCLES_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
fi
cd $ROOT
