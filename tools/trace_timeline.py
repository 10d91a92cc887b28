"""per-queue busy time / gaps / top kernels per step from a rocprofv3 kernel_trace.csv (steady-state tail of the run)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
# steady state: last `steps` occurrences of the optimizer's last kernel mark step ends; use the cell-kernel launches as step markers
marks = [r['s'] for r in rows if 'fps_cell_kernel' in r['Kernel_Name']]
t0, t1 = marks[-steps - 1], marks[-1]
sel = [r for r in rows if t0 <= r['s'] < t1]
print("window %.3f ms for %d steps -> %.3f ms/step" % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps))
byq = collections.defaultdict(list)
for r in sel:
    byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items()):
    rs.sort(key=lambda r: r['s'])
    busy = sum(r['e'] - r['s'] for r in rs)
    gaps = [b['s'] - a['e'] for a, b in zip(rs, rs[1:]) if b['s'] > a['e']]
    small = [g for g in gaps if g < 50_000]
    print("queue %s: %5.1f kernels/step, busy %.3f ms/step, gaps<50us: %.3f ms/step (n=%.0f/step, avg %.1f us)" % (
        q, len(rs) / steps, busy / 1e6 / steps, sum(small) / 1e6 / steps, len(small) / steps, (sum(small) / max(1, len(small))) / 1e3))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][0] += r['e'] - r['s']; agg[k][1] += 1
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
        print("     %-62s %5.1f/step  %7.3f ms/step  avg %7.1f us" % (k, n / steps, t / 1e6 / steps, t / n / 1e3))
