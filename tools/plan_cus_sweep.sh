#!/bin/bash
# full-step and layers-only ms/step for several GSPN_PLAN_CUS (rebuilds the library each time; run on the GPU box)
for c in ${@:-208 224 240 256}; do
  touch gspn_amd/csrc/common.h
  GSPN_EXTRA_HIPCC_FLAGS="-DGSPN_PLAN_CUS=$c" python -m gspn_amd.build > /dev/null 2>&1 || { echo "build failed for $c"; continue; }
  f=$(python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
  l=$(GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
  echo "PLAN_CUS $c: full $f ms  layers-only $l ms"
done
