#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 32768 65536; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_fps_${n}_$c -o p -- python tools/fps_only.py 5 $n > gpurun_out/pmc_fps_${n}_$c.log 2>&1
  done
  python tools/pmc_summary.py gpurun_out/pmc_fps_${n}_FETCH_SIZE gpurun_out/pmc_fps_${n}_WRITE_SIZE gpurun_out/r02_fps_pmc_$n.json | tail -3
done
