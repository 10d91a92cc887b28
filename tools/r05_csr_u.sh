#!/bin/bash
# csr_gather16: 8 vs 16 row loads in flight (GSPN_CSR_U1), on U and on the room scenes S (skewed list lengths)
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))"; }
for k in U S; do for lib in "" gspn_amd/lib/libgspn_hip_csru16.so; do
  for r in 1 2; do echo "data $k lib ${lib:-default}: $(GSPN_HIP_LIB=$lib python bench.py --data $k --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | ms)   layers only: $(GSPN_HIP_LIB=$lib GSPN_BENCH_LAYERS_ONLY=1 python bench.py --data $k --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | ms)"; done
done; done
