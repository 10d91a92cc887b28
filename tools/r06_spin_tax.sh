#!/bin/bash
# what a RESIDENT kernel on a side queue costs the captured layers by itself (tools/spin_probe.hip through GSPN_BENCH_SIDE=spin:which:blocks:threads:amount),
# one launch per step, each shorter than a step.  10 = hold_vgpr (sleeps, holds all vector registers of its CU: nobody co-resides), 5 = loop_valu_fullregs (busy VALU, all
# registers), 1 = spin_sleep (sleeps, small footprint: layer workgroups can share its CU), 3 = loop_valu (busy, small footprint)
run() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
echo "layers only                                     $(run A=1)"
for spec in 10:1:1024:120000 10:8:1024:120000 10:32:1024:120000 5:1:1024:3500 5:8:1024:3500 1:1:1024:120000 1:8:1024:120000 3:1:1024:3500 3:8:256:3500; do
  echo "spin $spec   $(run GSPN_BENCH_SIDE=spin:$spec)"
done
