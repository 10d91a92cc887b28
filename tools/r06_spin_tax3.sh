#!/bin/bash
# the sleeping / holding probes again with amounts in SHADER-clock cycles (clock64 = s_memtime runs at the shader clock, ~2.4 GHz: 2.9 M cycles = 1.2 ms)
run() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
echo "layers only                                     $(run A=1)"
for spec in 10:1:1024:2900000 10:8:1024:2900000 10:32:1024:2900000 1:1:1024:2900000 1:8:1024:2900000 9:8:1024:2900000 0:1:1024:2900000 0:1:64:2900000 2:1:1024:2900000; do
  echo "spin $spec   $(run GSPN_BENCH_SIDE=spin:$spec)"
done
