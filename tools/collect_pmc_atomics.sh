#!/bin/bash
# Atomic-request counters of the training kernels (run on the GPU box from the repo root):
#   bash tools/collect_pmc_atomics.sh <tag>   -> gpurun_out/<tag>_pmc_train_atomics.json
# One rocprofv3 --pmc pass per batch size over tools/train_probe.py (default method, 4096 and 65536 random rays: the set of
# cell-major levels depends on the batch size), never combined with API traces.
set -e
TAG=${1:-r02}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for R in 4096 65536; do
  O=$ROOT/gpurun_out/pmc_atomics_${TAG}_$R
  rm -rf $O; mkdir -p $O
  TRAIN_RAYS=$R rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O -- python3 $ROOT/tools/train_probe.py > $O/run.log 2>&1
done
cd $ROOT
python3 - "$ROOT/gpurun_out" "$TAG" <<'PY'
import csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
out = {"command": "TRAIN_RAYS=<rays> rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum -- python3 tools/train_probe.py  (default method: 48 field samples and 256 + 96 proposal samples per ray)",
       "commit": os.environ.get("CN_PROFILE_COMMIT", ""), "rays": {}}
for rays in (4096, 65536):
    acc = {}
    for f in glob.glob(os.path.join(root, f"pmc_atomics_{tag}_{rays}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc.setdefault((name, row["Counter_Name"]), [0.0, set()])
            a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
    k = {}
    for (name, counter), (total, ids) in sorted(acc.items()):
        if total > 0:
            k.setdefault(name, {})[counter] = {"launches": len(ids), "avg_per_launch": total / len(ids)}
    out["rays"][str(rays)] = k
    for name, v in k.items():
        print(rays, name, {c: round(x["avg_per_launch"]) for c, x in v.items()})
json.dump(out, open(os.path.join(root, f"{tag}_pmc_train_atomics.json"), "w"), indent=1)
PY
