#!/bin/bash
# everything profiles/ holds for the current round, from the current tree (run on the GPU box; results land in gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r02}
# 1. the bench line the driver would record
python bench.py 2> gpurun_out/${R}_bench_default.log | tail -1 > gpurun_out/${R}_bench_line.json
# 2. per-kernel table of the same command
rm -rf gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -o k -- python bench.py --steps 40 --no-cpu-baseline --no-extra > gpurun_out/${R}_bench_under_rocprof.log 2>&1
cp $(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
# 3. one captured step, kernel by kernel
bash tools/trace_layers.sh > /dev/null; cp gpurun_out/layers_timeline.txt gpurun_out/${R}_layers_timeline.txt
# 4. counters (separate passes)
bash tools/pmc_fps.sh > gpurun_out/pmc_fps.out 2>&1
bash tools/pmc_f2.sh > gpurun_out/pmc_f2.out 2>&1
bash tools/pmc_sq.sh > gpurun_out/pmc_sq.out 2>&1
ls -la gpurun_out/${R}_* | head -20
