#!/bin/bash
# Kernel traces of the two method specifications whose field shape the fused kernels are not built for (run on the GPU box from
# the repo root):   bash tools/generic_shapes_profile.sh [render|train|both]   -> gpurun_out/generic_<mode>_<method>/ + a summary
# on stdout (copied to profiles/ by hand).  render: one 65 536-ray inference call; train: seven iterations at 8 192 rays.
MODE=${1:-both}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for M in fruit_nerf_method_big fruit_nerf_method_huge; do
  O=$ROOT/gpurun_out/generic_${MODE}_$M
  rm -rf $O; mkdir -p $O
  PROBE=$MODE METHOD=$M rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/tools/big_shape_probe.py > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $M ($MODE)"; grep "ms_per" $O/run.log
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {float(r['Percentage']):5.1f} %")
PY
done
