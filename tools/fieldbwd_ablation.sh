#!/bin/bash
# Where the field backward's time goes at the three training workloads of bench.py: the iteration time of tools/train_probe.py
# with parts of cn_field_backward switched off (CN_DEBUG_SKIP: 1 hash scatter, 2 embedding atomics, 4 weight-gradient products,
# 32 semantic branch, 64 forward gathers).  Eager launches for every run (CN_TRAIN_GRAPH=0) so that the deltas are kernel time.
#   tools/fieldbwd_ablation.sh > gpurun_out/fieldbwd_ablation.txt
export CN_TRAIN_GRAPH=0 WARM=3 ITERS=12
for cfg in "4096 48" "65536 48" "65536 192"; do
  set -- $cfg
  for skip in 0 1 4 5 7 64 65 71 103; do
    TRAIN_RAYS=$1 TRAIN_FIELD_SAMPLES=$2 CN_DEBUG_SKIP=$skip python3 tools/train_probe.py | sed "s/^/S=$2 /"
  done
done
