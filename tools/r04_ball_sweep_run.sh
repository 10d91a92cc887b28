#!/bin/bash
for pf in 2048 4096 6144 8192; do
  echo "== prefix $pf (4 waves x 4 queries)"; GSPN_BALL_PREFIX=$pf GSPN_HIP_LIB=$GRAFT_REPO_ROOT/gspn_amd/lib/libgspn_hip_bq_4_4.so python tools/ball_bench.py U S D 2>&1 | grep "n=32768" | sed 's/(.*scan)//'
done
