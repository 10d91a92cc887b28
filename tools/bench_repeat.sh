#!/bin/bash
# the default bench N times (spread of ms/step), optional env assignments as arguments
for i in 1 2 3 4 5; do
  env "$@" python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],3))"
done
