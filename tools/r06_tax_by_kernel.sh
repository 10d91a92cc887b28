#!/bin/bash
# where the geometry tax lands: per-kernel average durations (rocprofv3 --kernel-trace --stats) of the layer kernels in the captured-layers-only run and in the full step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in layers full; do
  rm -rf gpurun_out/tk_$tag
  if [ $tag = layers ]; then E="GSPN_BENCH_LAYERS_ONLY=1"; else E="A=1"; fi
  (cd /tmp; env $E rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tk_$tag -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tk_$tag.log 2>&1)
  cp $(find gpurun_out/tk_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r06_tk_$tag.csv; rm -rf gpurun_out/tk_$tag
done
python - <<'PY'
import csv
def load(p):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs'])) for r in csv.DictReader(open(p))}
a, b = load('gpurun_out/r06_tk_layers.csv'), load('gpurun_out/r06_tk_full.csv')
rows = []
for k, (n, t) in a.items():
    if k in b and n >= 60:
        rows.append((b[k][1] * n - t * n, k, n, t, b[k][1]))
rows.sort(reverse=True)
steps = 70.0
print("kernel (calls in the layers-only run)                                             layers-only us   full step us   ratio   extra us per step")
tot = 0
for d, k, n, t, tb in rows:
    tot += d
    print("%-84s %8.1f %12.1f %9.3f %10.1f" % ((k[:72] + " (%d)" % n), t / 1e3, tb / 1e3, tb / t, d / 1e3 / steps))
print("sum of the extra kernel time: %.1f us per step" % (tot / 1e3 / steps))
PY
