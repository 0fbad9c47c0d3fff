#!/bin/bash
# Where does field_backward_general_kernel's tile go?  Counters of the _big method's training iteration (run on the GPU box from the
# repo root):   bash tools/collect_pmc_genbwd.sh <tag>  -> gpurun_out/<tag>_genbwd_pmc.json  (per launch of the kernel)
# One rocprofv3 run per counter group (never combined with API traces), each under its own timeout.
TAG=${1:-r05}
ROOT=$(pwd)
O=$ROOT/gpurun_out/genbwd_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  local d=$1; shift
  rm -rf $O/$d; mkdir -p $O/$d
  PROBE=train timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$d -- python3 $ROOT/tools/big_shape_probe.py > $O/$d.log 2>&1 || echo "pass $d failed" >> $O/failed.log
  echo "pass $d done"
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
cd $ROOT
python3 - "$O" "$TAG" <<'PY'
import csv, glob, json, os, sys
o, tag = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(o, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "field_backward_general_kernel" not in row["Kernel_Name"]:
            continue
        a = acc.setdefault(row["Counter_Name"], [0.0, set()])
        a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
out = {"kernel": "cn::gb::field_backward_general_kernel", "workload": "fruit_nerf_method_big, 8192 rays x 128 samples (tools/big_shape_probe.py, PROBE=train)",
       "per_launch": {k: v[0] / max(len(v[1]), 1) for k, v in sorted(acc.items())}}
json.dump(out, open(os.path.join(os.path.dirname(o), f"{tag}_genbwd_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
