run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), round(d['ms_per_step'],3))"; }
for cfg in "8 1" "4 1" "2 1"; do
  set -- $cfg
  touch cropnerf-a-neural-radiance-field-based-framework_amd/csrc/render_split.hpp
  CN_EXTRA_HIPCC_FLAGS="-DCN_SPLIT_G=$1 -DCN_SPLIT_MPG=$2" python cropnerf-a-neural-radiance-field-based-framework_amd/build.py > /dev/null 2>&1 || echo build failed
  run "G=$1 MPG=$2"
  for st in 2 3 6; do CN_STRIPES_PER_XCD=$st run "   stripes=$st"; done
done
