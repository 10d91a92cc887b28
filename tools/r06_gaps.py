"""per step of the LAYERS' queue (the queue that runs adam_flat*): number of kernels, busy time, sum of gaps between consecutive kernels, the gap in front of the
first kernel of the step and in front of Adam; median over the steps of a rocprofv3 kernel trace"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
adam = [r for r in rows if r["Kernel_Name"].startswith("adam_flat")]
q = adam[0][qkey] if qkey else None
rs = sorted([r for r in rows if (not qkey or r[qkey] == q)], key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rs) if r["Kernel_Name"].startswith("adam_flat")]
steps = []
for a, b in zip(ends[2:-1], ends[3:]):
    seg = rs[a + 1:b + 1]
    t_prev = int(rs[a]["End_Timestamp"])
    busy = gaps = 0
    first_gap = int(seg[0]["Start_Timestamp"]) - t_prev
    adam_gap = int(seg[-1]["Start_Timestamp"]) - int(seg[-2]["End_Timestamp"])
    big = 0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = max(0, s - t_prev)
        gaps += g; busy += e - s; t_prev = max(t_prev, e)
        if g > 3000: big += 1
    steps.append((len(seg), busy / 1e3, gaps / 1e3, first_gap / 1e3, adam_gap / 1e3, big, (int(seg[-1]["End_Timestamp"]) - int(rs[a]["End_Timestamp"])) / 1e3))
med = lambda i: st.median(s[i] for s in steps)
print("%-7s steps %d: kernels %d  busy %.1f us  gaps %.1f us (first %.1f, before Adam %.1f, gaps > 3 us: %.0f per step)  span %.1f us" % (
    sys.argv[2], len(steps), med(0), med(1), med(2), med(3), med(4), med(5), med(6)))
