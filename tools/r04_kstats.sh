#!/bin/bash
# per-kernel stats of the bench command on a cloud kind: tools/r04_kstats.sh S  -> gpurun_out/r04_kstats_S.csv (+ top rows printed)
K=${1:-U}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kstats_$K
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kstats_$K -o k -- python $GRAFT_REPO_ROOT/bench.py --data $K --steps 40 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/kstats_$K.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/kstats_$K -name "*kernel_stats.csv" | head -1) gpurun_out/r04_kstats_$K.csv
rm -rf gpurun_out/kstats_$K
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_kstats_$K.csv")))
geo = ("fps", "ball", "three_nn", "csr", "gather_point", "sa_rel", "multi_copy", "cell", "bin", "weights", "prepass", "morton", "order")
print("kind $K: geometry-side kernels (per call, calls, total ms)")
tot = 0.0
for r in rows:
    n = r["Name"].split("(")[0]
    if any(g in n for g in geo):
        t = float(r["TotalDurationNs"]) / 1e6
        tot += t
        print("  %-60s %9.1f us x %5s = %8.2f ms" % (n[:60], float(r["AverageNs"]) / 1e3, r["Calls"], t))
print("  geometry total %.2f ms" % tot)
PY
