#!/bin/bash
# inter-kernel gaps of the layers' stream inside a step: captured layers alone against the full step (geometry on the side queues)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in layers full; do
  rm -rf gpurun_out/gp_$tag
  if [ $tag = layers ]; then E="GSPN_BENCH_LAYERS_ONLY=1"; else E="A=1"; fi
  (cd /tmp; env $E rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gp_$tag -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/gp_$tag.log 2>&1)
  python tools/r06_gaps.py $(find gpurun_out/gp_$tag -name "*kernel_trace.csv" | head -1) $tag
  rm -rf gpurun_out/gp_$tag
done
