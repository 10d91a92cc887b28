#!/bin/bash
# kernel-by-kernel timeline of one captured step (layers only): name, duration, gap to the previous kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ltrace
GSPN_BENCH_LAYERS_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ltrace -- python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > gpurun_out/ltrace.log 2>&1
python tools/trace_layers.py $(find gpurun_out/ltrace -name "*kernel_trace.csv" | head -1) > gpurun_out/layers_timeline.txt
tail -5 gpurun_out/layers_timeline.txt
