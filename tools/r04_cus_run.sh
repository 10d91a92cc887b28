#!/bin/bash
ms() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3))"; }
for rep in 1 2; do
for c in 208 224 232 240; do
  lib=$GRAFT_REPO_ROOT/gspn_amd/lib/libgspn_hip_cus$c.so; [ $c = 224 ] && lib=$GRAFT_REPO_ROOT/gspn_amd/lib/libgspn_hip.so
  echo "PLAN_CUS $c: full $(GSPN_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extra --steps 200 2>/dev/null | tail -1 | ms)  layers $(GSPN_HIP_LIB=$lib GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra --steps 200 2>/dev/null | tail -1 | ms)"
done; done
