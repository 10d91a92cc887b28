#!/bin/bash
# layers-only and full step under a list of env settings: tools/r04_ab.sh "A=1" "GSPN_X=0" ...
ms() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3))"; }
for v in "$@"; do
  echo "$v   layers $(env $v GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra --steps 200 2>/dev/null | tail -1 | ms)   full $(env $v python bench.py --no-cpu-baseline --no-extra --steps 200 2>/dev/null | tail -1 | ms)"
done
