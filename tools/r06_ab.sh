#!/bin/bash
# A/B of one environment switch on the captured layers alone and on the full step (ms per step, 100 steps, two runs each)
# usage: tools/r06_ab.sh VAR=off_value VAR=on_value
run() { env "$@" python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))"; }
for v in "$@"; do
  for rep in 1 2; do
    echo "$v layers-only $(run $v GSPN_BENCH_LAYERS_ONLY=1)   full $(run $v)"
  done
done
