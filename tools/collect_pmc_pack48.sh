#!/bin/bash
# SQ counters of the 48-sample field pass under both schedules (run on the GPU box from the repo root):
#   bash tools/collect_pmc_pack48.sh <tag>  -> gpurun_out/<tag>_pack48_sq.json
# CN_SPLIT_PACK=0: one ray per two half-steps (every fourth 16-sample column tile empty); default: two rays per three.
TAG=${1:-r04}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pack48_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in 0 1; do
  d=$O/pack$P; rm -rf $d; mkdir -p $d
  CN_SPLIT_PACK=$P timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $d -- python3 $ROOT/tools/pmc_workloads.py render48 > $d/run.log 2>&1
  echo "pack=$P rc=$? $(grep PMC_UNITS $d/run.log | tail -1)"
  d=$O/trace$P; rm -rf $d; mkdir -p $d
  CN_SPLIT_PACK=$P timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $ROOT/tools/pmc_workloads.py render48 > $d/run.log 2>&1
done
cd $ROOT
python3 - "$O" "$TAG" <<'PY'
import csv, glob, json, os, sys
o, tag = sys.argv[1], sys.argv[2]
out = {"command": "CN_SPLIT_PACK=<0|1> rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- "
                  "python3 tools/pmc_workloads.py render48   (10 launches of 65 536 rays x 48 samples; kernel-trace in a separate run)",
       "commit": os.environ.get("CN_PROFILE_COMMIT", ""), "schedules": {}}
for p, name in (("0", "one_ray_per_two_half_steps"), ("1", "two_rays_per_three_half_steps")):
    acc = {}
    for f in glob.glob(os.path.join(o, f"pack{p}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_split_kernel" not in row["Kernel_Name"]:
                continue
            a = acc.setdefault(row["Counter_Name"], [0.0, set(), row["Kernel_Name"]])
            a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
    e = {k: v[0] / max(len(v[1]), 1) for k, v in acc.items()}
    e["kernel"] = next(iter(acc.values()))[2] if acc else None
    for f in glob.glob(os.path.join(o, f"trace{p}", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_split_kernel" in row["Name"]:
                e["kernel_trace_avg_ms"] = float(row["AverageNs"]) / 1e6
                e["kernel_trace_launches"] = int(row["Calls"])
    out["schedules"][name] = e
a, b = out["schedules"].get("one_ray_per_two_half_steps", {}), out["schedules"].get("two_rays_per_three_half_steps", {})
if a.get("SQ_VALU_MFMA_BUSY_CYCLES") and b.get("SQ_VALU_MFMA_BUSY_CYCLES"):
    out["mfma_busy_ratio"] = b["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_VALU_MFMA_BUSY_CYCLES"]
    out["valu_inst_ratio"] = b["SQ_ACTIVE_INST_VALU"] / a["SQ_ACTIVE_INST_VALU"]
json.dump(out, open(os.path.join(os.path.dirname(o), f"{tag}_pack48_sq.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
