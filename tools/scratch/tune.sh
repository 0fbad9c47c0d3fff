run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), round(d['ms_per_step'],3))"; }
for v in 0 1; do
  touch cropnerf-a-neural-radiance-field-based-framework_amd/csrc/render_split.hpp
  CN_EXTRA_HIPCC_FLAGS="-DCN_SPLIT_BASE_IN_GATHER=$v" python cropnerf-a-neural-radiance-field-based-framework_amd/build.py > /dev/null 2>&1 || echo build failed
  run "base_in_gather=$v"
done
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
