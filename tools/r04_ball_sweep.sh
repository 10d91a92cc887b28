#!/bin/bash
# variants of the continuation kernel's blocking (waves per workgroup x queries per wave): builds variant libraries, times U and S
for cfg in "8 8" "8 4" "8 2" "16 4" "16 2" "4 4"; do
  set -- $cfg
  python - <<PY
from gspn_amd import build
build.build(variant="bq_$1_$2", extra_flags=["-DBQM_WAVES=$1", "-DBQM_QW=$2"])
PY
done
