#!/bin/bash
# A/B of the short-layer kernels (r05): the same launches with the path off and on, and forced shapes
W=${1:-fwd}
echo "== short kernels off (round-4 kernels)"; GSPN_FWD_SHORT=0 GSPN_BWD_SHORT=0 GSPN_WGRAD_SHORT=0 python tools/short_bench.py $W 2>&1 | grep -v amdgpu.ids
echo "== default (+ GSPN_WGRAD_SHORT=1)"; GSPN_WGRAD_SHORT=1 python tools/short_bench.py $W 2>&1 | grep -v amdgpu.ids
if [ "$W" = "wgrad" ]; then
  for wgs in 224 672 896; do echo "== GSPN_WGRAD_SHORT_WGS=$wgs"; GSPN_WGRAD_SHORT=1 GSPN_WGRAD_SHORT_WGS=$wgs python tools/short_bench.py $W 2>&1 | grep -v amdgpu.ids; done
fi
