#!/bin/bash
# memory-side bytes per step with and without the fused front ends (gathered / pre-aggregated first layers of SA and FP modules) and the
# pool epilogue (2 separate --pmc passes each: one counter per run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STEPS=4
for v in fused unfused; do
  if [ $v = unfused ]; then export GSPN_FUSE_SA_FRONT=0 GSPN_FUSE_POOL32=0 GSPN_PREAGG=0 GSPN_FUSE_FP_FRONT=0; else unset GSPN_FUSE_SA_FRONT GSPN_FUSE_POOL32 GSPN_PREAGG GSPN_FUSE_FP_FRONT; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${v}_$c -o p -- python bench.py --steps $STEPS --warmup 0 --no-cpu-baseline --no-extra --no-graph --no-overlap > gpurun_out/pmc_${v}_$c.log 2>&1
  done
  python tools/pmc_step_bytes.py gpurun_out/pmc_${v}_FETCH_SIZE gpurun_out/pmc_${v}_WRITE_SIZE $STEPS gpurun_out/r02_step_bytes_$v.json
done
