#!/bin/bash
# Kernel-trace stats of the training iteration (run on the GPU box from the repo root):
#   bash tools/train_profile.sh <tag>   -> gpurun_out/train_<tag>_{4096,65536}/ ... *_kernel_stats.csv
TAG=${1:-r02}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for R in 4096 65536; do
  O=$ROOT/gpurun_out/train_${TAG}_$R
  mkdir -p $O
  TRAIN_RAYS=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/tools/train_probe.py > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  cp "$f" $ROOT/gpurun_out/${TAG}_train_${R}_kernel_stats.csv
  echo "== $R rays: $(grep ms/iter $O/run.log | tail -1)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {float(r['Percentage']):5.1f} %")
PY
done
