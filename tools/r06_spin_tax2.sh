#!/bin/bash
run() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
echo "layers only                                     $(run A=1)"
for spec in 5:1:1024:500 5:1:1024:1000 5:1:1024:2000 5:8:1024:1000 5:8:1024:2000 5:32:1024:2000 3:1:1024:2000 12:1:1024:2000 4:1:1024:2000; do
  echo "spin $spec   $(run GSPN_BENCH_SIDE=spin:$spec)"
done
