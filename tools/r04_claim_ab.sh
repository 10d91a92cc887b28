#!/bin/bash
ms() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3))"; }
full() { env "$@" python bench.py --no-cpu-baseline --no-extra --steps 200 2>/dev/null | tail -1 | ms; }
for rep in 1 2 3; do
for c in 0 2 4 6; do echo "claim=$c  $(full GSPN_GEOM_CLAIM_LDS=$c)"; done
done
