#!/bin/bash
# schedule variants of the geometry prefetch: ms per step (mean median) and the replay span diagnostic, 200 steps
run() { env "$@" GSPN_BENCH_GAPS=1 python bench.py --no-cpu-baseline --steps 200 2>&1 | grep -a "replay span\|^{" | python -c "
import sys, json
span = ''
for l in sys.stdin:
    if l.startswith('replay span'): span = l.split('ms:')[1].strip()[:70] + ' | ' + l.split(';')[1].strip()
    elif l.startswith('{'): d = json.loads(l); print(round(d['ms_per_step'],4), round(d['median_ms_per_step'],4), '|', span)
"; }
echo "paired GROUP=2 (default)        $(run A=1)"
echo "paired GROUP=2 (default)        $(run A=1)"
echo "paired GROUP=3                  $(run GSPN_BENCH_GROUP=3)"
echo "paired GROUP=4                  $(run GSPN_BENCH_GROUP=4)"
echo "unpaired DEPTH=2 NB=3           $(run GSPN_BENCH_PAIRED=0)"
echo "unpaired DEPTH=3 NB=4           $(run GSPN_BENCH_PAIRED=0 GSPN_BENCH_DEPTH=3 GSPN_BENCH_NB=4)"
echo "unpaired DEPTH=4 NB=5           $(run GSPN_BENCH_PAIRED=0 GSPN_BENCH_DEPTH=4 GSPN_BENCH_NB=5)"
