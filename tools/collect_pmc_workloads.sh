#!/bin/bash
# Memory-side counters of bench.py's SECONDARY workloads (run on the GPU box from the repo root):
#   bash tools/collect_pmc_workloads.sh <tag>   -> gpurun_out/pmcw_<tag>/<workload>_<group>/ ; then
#   python3 tools/summarise_pmc_workloads.py <tag>  -> profiles/<tag>_pmc_workloads.json (+ <tag>_pmc_train_atomics.json)
# One rocprofv3 --pmc run per workload and counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass), the program directly
# after `--`, never combined with API traces; every pass under its own timeout.
TAG=${1:-r04}
ROOT=$(pwd)
O=$ROOT/gpurun_out/pmcw_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counter group, program arguments / environment come from the caller's environment
  local name=$1 group=$2; shift 2
  local d=$O/${name}_$(echo $group | cut -d' ' -f1)
  rm -rf $d; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $d -- python3 $ROOT/tools/pmc_workloads.py "$@" > $d/run.log 2>&1
  echo "$name [$group]: rc=$? $(grep PMC_UNITS $d/run.log | tail -1)"
}
for W in ${PMC_WORKLOADS:-eval_image projection export_c4 dense_export proposal}; do
  pass $W "FETCH_SIZE" $W
  pass $W "WRITE_SIZE" $W
done
[ -n "$PMC_SKIP_TRAIN" ] || for CFG in "4096 48" "65536 48" "65536 192"; do
  set -- $CFG
  export TRAIN_RAYS=$1 TRAIN_FIELD_SAMPLES=$2
  N=train_${1}x${2}
  pass $N "FETCH_SIZE" train
  pass $N "WRITE_SIZE" train
  pass $N "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" train
done
echo done
