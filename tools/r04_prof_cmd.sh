#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python command: tools/r04_prof_cmd.sh <tag> <script> [args...] -> prints the top kernels
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o k -- python $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${TOPN:-14}]:
    print("%-70s calls %6s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_$TAG
