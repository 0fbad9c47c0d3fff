#!/bin/bash
# Counter passes for the dominant render kernel (run on the GPU box from the repo root):
#   bash tools/collect_pmc.sh <tag>      -> gpurun_out/pmc_<tag>/{fetch,write,l2,sq,tcp}/ + kernel-trace stats
# One rocprofv3 run per counter group, each under its own timeout (a pass that asks for more counters than a block has hangs) (PMC passes serialise kernels; never combined with API traces), then
# tools/summarise_pmc.py folds the CSVs into profiles/<tag>_pmc_render_fused.json and <tag>_kernel_stats.csv.
set -e
TAG=${1:-r01}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
B="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $B > $O/fetch.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $B > $O/write.log 2>&1
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/l2 -- python3 $B > $O/l2.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $B > $O/sq.log 2>&1
timeout -k 10 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr --output-format csv -d $O/tcp -- python3 $B > $O/tcp.log 2>&1
cd $ROOT
python3 tools/summarise_pmc.py $TAG || true
