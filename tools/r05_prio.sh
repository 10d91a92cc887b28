#!/bin/bash
# stream priorities of the layers' stream / the geometry streams (HIP: -1 high, 0 normal, 1 low): ms per step (mean, median)
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))"; }
for e in "X=0" "GSPN_BENCH_MAIN_PRIO=-1" "GSPN_BENCH_GEO_PRIO=1" "GSPN_BENCH_MAIN_PRIO=-1 GSPN_BENCH_GEO_PRIO=1"; do
  for r in 1 2; do echo "$e: $(env $e python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | ms)"; done
done
