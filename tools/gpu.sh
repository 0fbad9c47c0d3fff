#!/bin/bash
# Build the HIP library (a stale .so would travel to the GPU box as it is), then run a command on the MI355X box.
#   tools/gpu.sh [--timeout N] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python cropnerf-a-neural-radiance-field-based-framework_amd/build.py > /tmp/cn_build.log 2>&1 || { tail -30 /tmp/cn_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
