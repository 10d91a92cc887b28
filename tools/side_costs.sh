#!/bin/bash
# what each part of the geometry costs the captured layers when it runs beside them (bench diagnostics, ms per step over 100 steps)
run() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
echo "layers only        $(run A=1)"
for s in fps0 inv nn small rest "fps0 rest"; do echo "side '$s'   $(run GSPN_BENCH_SIDE="$s")"; done
echo "full step          $(python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")"
