#!/bin/bash
# Atomic-request counters of the training kernels (run on the GPU box from the repo root):
#   bash tools/collect_pmc_atomics.sh <tag>   -> profiles-style JSON in gpurun_out/<tag>_pmc_train_atomics.json
# One rocprofv3 --pmc pass over tools/train_probe.py (default method, 4096 random rays), never combined with API traces.
set -e
TAG=${1:-r02}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmc_atomics_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O -- python3 $ROOT/tools/train_probe.py > $O/run.log 2>&1
cd $ROOT
python3 - "$O" "$TAG" <<'PY'
import csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc.setdefault((name, row["Counter_Name"]), [0.0, set()])
        a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
out = {"command": "rocprofv3 --pmc TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_ATOMIC_sum -- python3 tools/train_probe.py  (default method, 4096 rays = 196 608 field samples, 1 441 792 proposal samples per iteration)",
       "commit": os.environ.get("CN_PROFILE_COMMIT", ""), "kernels": {}}
for (name, counter), (total, ids) in sorted(acc.items()):
    if total > 0:
        out["kernels"].setdefault(name, {})[counter] = {"launches": len(ids), "avg_per_launch": total / len(ids)}
json.dump(out, open(os.path.join(os.path.dirname(src), f"{tag}_pmc_train_atomics.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, {c: round(x["avg_per_launch"]) for c, x in v.items()})
PY
