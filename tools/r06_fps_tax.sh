#!/bin/bash
# how the tax of FPS of SA level 1 on the captured layers scales with the number of scenes (= CUs held) per launch, one launch per step
run() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
echo "layers only        $(run A=1)"
for n in 1 2 4 8 16 32; do echo "FPS of $n scenes per step beside them   $(run GSPN_BENCH_SIDE="fpsn:$n")"; done
