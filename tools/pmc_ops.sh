#!/bin/bash
# memory-side bytes per launch of the stand-alone ops of bench.py's roofline_ops leg -> gpurun_out/r04_ops_pmc.json (copy to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_ops_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_ops_$c -o p -- python tools/ops_only.py > gpurun_out/pmc_ops_$c.log 2>&1
done
python tools/pmc_ops_summary.py gpurun_out/pmc_ops_FETCH_SIZE gpurun_out/pmc_ops_WRITE_SIZE gpurun_out/pmc_ops_FETCH_SIZE.log gpurun_out/r04_ops_pmc.json
rm -rf gpurun_out/pmc_ops_FETCH_SIZE gpurun_out/pmc_ops_WRITE_SIZE
