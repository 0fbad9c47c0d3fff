#!/bin/bash
# Kernel trace + counter passes for ONE render variant of bench.py (run on the GPU box from the repo root):
#   bash tools/collect_pmc_variant.sh <tag> <variant>      e.g.  r03_tcnn_f16 tcnn_f16
# -> gpurun_out/pmc_<tag>/{stats,fetch,write,l2,sq,sq2,tcp,tlb}/ ; tools/summarise_pmc.py <tag> folds them into profiles/.
# One rocprofv3 run per counter group (PMC passes serialise kernels; never combined with API traces).
set -e
TAG=${1:-r03_tcnn_f16}
VAR=${2:-tcnn_f16}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
B="$ROOT/bench.py --variant $VAR --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ROOT/bench.py --variant $VAR --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $B > $O/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/l2 -- python3 $B > $O/l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $B > $O/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- python3 $B > $O/sq2.log 2>&1 || true
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr --output-format csv -d $O/tcp -- python3 $B > $O/tcp.log 2>&1
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $O/tlb -- python3 $B > $O/tlb.log 2>&1 || true
# where a gather-bound variant waits: texture-address / L1 / L2 request path
rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $O/ta -- python3 $B > $O/ta.log 2>&1 || true
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TD_TCP_STALL_CYCLES_sum --output-format csv -d $O/tcp2 -- python3 $B > $O/tcp2.log 2>&1 || true
rocprofv3 --pmc TCC_TAG_STALL_sum TCC_BUSY_sum TCC_READ_sum --output-format csv -d $O/tcc2 -- python3 $B > $O/tcc2.log 2>&1 || true
cd $ROOT
python3 tools/summarise_pmc.py $TAG pmc_variant
