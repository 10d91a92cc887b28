#!/bin/bash
# is the geometry tax co-residency?  (i) busy neighbour with / without the CU's whole LDS, (ii) geometry kernels claiming the LDS
ms() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), 'fps_ms', round(r['roofline']['avg_launch_ms'],3))"; }
lay() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | ms; }
full() { env "$@" python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | ms; }
echo "layers only                         $(lay A=1)"
echo "layers + 8 x loop_valu 1024 thr     $(lay GSPN_BENCH_SIDE=spin:3:8:1024:6000)"
echo "layers + 8 x loop_valu + 160KB LDS  $(lay GSPN_BENCH_SIDE=spin:12:8:1024:6000)"
echo "layers + 1 x loop_valu 1024 thr     $(lay GSPN_BENCH_SIDE=spin:3:1:1024:6000)"
echo "layers + 1 x loop_valu + 160KB LDS  $(lay GSPN_BENCH_SIDE=spin:12:1:1024:6000)"
echo "layers + fps0                       $(lay GSPN_BENCH_SIDE=fps0)"
echo "layers + fps0, claim LDS            $(lay GSPN_BENCH_SIDE=fps0 GSPN_GEOM_CLAIM_LDS=1)"
echo "full step                           $(full A=1)"
echo "full step, fps_cell claims LDS      $(full GSPN_GEOM_CLAIM_LDS=1)"
echo "full step, cell+small claim         $(full GSPN_GEOM_CLAIM_LDS=3)"
echo "full step, cell+small+csr claim     $(full GSPN_GEOM_CLAIM_LDS=7)"
echo "full step (repeat)                  $(full A=1)"
