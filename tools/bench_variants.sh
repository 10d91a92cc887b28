#!/bin/bash
# layers-only and full-step bench under a few switches (diagnostic)
for v in "" "GSPN_FUSE_POOL32=0" "GSPN_FUSE_SA_FRONT=0" "GSPN_FUSE_POOL32=0 GSPN_FUSE_SA_FRONT=0"; do
  echo "== $v"
  env $v GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('layers-only ms', round(r['ms_per_step'],3))"
  env $v python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('full ms', round(r['ms_per_step'],3), 'fps ms', round(r['roofline']['avg_launch_ms'],3))"
done
