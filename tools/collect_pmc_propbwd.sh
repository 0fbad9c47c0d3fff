#!/bin/bash
# SQ counters of cn_proposal_backward's kernel in the training iteration (run on the GPU box from the repo root):
#   bash tools/collect_pmc_propbwd.sh <tag>  -> gpurun_out/<tag>_propbwd_sq.json
# per launch and batch size (4 096 / 65 536 rays), both forms (CN_PROP_BWD=wave|tile)
TAG=${1:-r04}
ROOT=$(pwd)
O=$ROOT/gpurun_out/propbwd_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for R in 4096 65536; do for F in wave tile; do
  d=$O/${F}_$R; rm -rf $d; mkdir -p $d
  CN_TRAIN_GRAPH=0 CN_PROP_BWD=$F TRAIN_RAYS=$R WARM=2 ITERS=6 timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $d -- python3 $ROOT/tools/train_probe.py > $d/run.log 2>&1
  echo "$F $R rc=$? $(grep ms/iter $d/run.log | tail -1)"
done; done
cd $ROOT
python3 - "$O" "$TAG" <<'PY'
import csv, glob, json, os, sys
o, tag = sys.argv[1], sys.argv[2]
out = {"command": "CN_TRAIN_GRAPH=0 CN_PROP_BWD=<form> TRAIN_RAYS=<rays> rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
                  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/train_probe.py", "runs": {}}
for d in sorted(glob.glob(os.path.join(o, "*_*"))):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "proposal_backward" not in row["Kernel_Name"]:
                continue
            a = acc.setdefault(row["Counter_Name"], [0.0, set()])
            a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
    out["runs"][os.path.basename(d)] = {k: v[0] / max(len(v[1]), 1) for k, v in acc.items()}
json.dump(out, open(os.path.join(os.path.dirname(o), f"{tag}_propbwd_sq.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
