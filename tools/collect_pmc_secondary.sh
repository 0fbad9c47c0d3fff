#!/bin/bash
# Counter passes for the secondary kernels (run on the GPU box from the repo root):
#   bash tools/collect_pmc_secondary.sh <tag>   -> gpurun_out/pmc2_<tag>/{prop_*,train_*}/
# proposal-mode bench (proposal_sample_kernel) and the training probe (field / proposal backward, Adam).  One rocprofv3 run
# per counter group, never combined with API traces; tools/summarise_pmc_secondary.py folds them into profiles/.
set -e
TAG=${1:-r01}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmc2_$TAG
mkdir -p $O
P="$ROOT/bench.py --mode proposal --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
T="$ROOT/tools/train_probe.py"
cd /tmp && export TMPDIR=/tmp
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  N=$(echo $G | cut -d' ' -f1)
  rocprofv3 --pmc $G --output-format csv -d $O/prop_$N -- python3 $P > $O/prop_$N.log 2>&1
  rocprofv3 --pmc $G --output-format csv -d $O/train_$N -- python3 $T > $O/train_$N.log 2>&1
done
echo done
