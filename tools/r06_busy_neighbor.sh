#!/bin/bash
# which layer kernels does ONE busy, isolated workgroup on a side queue slow down?  per-kernel average durations (rocprofv3 --kernel-trace --stats) of the
# captured layers alone and beside loop_valu_fullregs (1 workgroup, 1024 threads, all vector registers of its CU: nobody co-resides), ~0.9 ms per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in alone busy sleep; do
  rm -rf gpurun_out/bn_$tag
  case $tag in alone) S="";; busy) S="spin:5:1:1024:1000";; sleep) S="spin:10:1:1024:2200000";; esac
  (cd /tmp; GSPN_BENCH_LAYERS_ONLY=1 GSPN_BENCH_SIDE="$S" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bn_$tag -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bn_$tag.log 2>&1)
  cp $(find gpurun_out/bn_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r06_bn_$tag.csv; rm -rf gpurun_out/bn_$tag
done
python - <<'PY'
import csv
def load(p):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs'])) for r in csv.DictReader(open(p))}
a, b, c = load('gpurun_out/r06_bn_alone.csv'), load('gpurun_out/r06_bn_busy.csv'), load('gpurun_out/r06_bn_sleep.csv')
rows = []
for k, (n, t) in a.items():
    if k in b and k in c and n >= 40:
        rows.append((n * t, k, n, t, b[k][1], c[k][1]))
rows.sort(reverse=True)
ta = sum(r[0] for r in rows); tb = sum(r[2] * r[4] for r in rows); tc = sum(r[2] * r[5] for r in rows)
print("kernel (calls)                                                              alone us   busy-neighbour us  ratio   sleeping-neighbour us  ratio")
for tot, k, n, t, tb_, tc_ in rows[:45]:
    print("%-80s %8.1f %12.1f %10.3f %14.1f %10.3f" % ((k[:70] + " (%d)" % n), t / 1e3, tb_ / 1e3, tb_ / t, tc_ / 1e3, tc_ / t))
print("sum over these kernels: alone %.1f ms, busy %.1f ms (%.3f), sleeping %.1f ms (%.3f)" % (ta / 1e6, tb / 1e6, tb / ta, tc / 1e6, tc / ta))
for nm, d in (("busy", b), ("sleep", c)):
    for k, v in d.items():
        if "loop_valu" in k or "hold_vgpr" in k: print(nm, k[:60], v)
PY
