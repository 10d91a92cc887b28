"""one step of the captured layers from a rocprofv3 kernel trace: per kernel its duration and the gap since the previous one ended"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step ends with the optimiser kernel
ends = [i for i, n in enumerate(names) if n.startswith("adam_flat")]
if len(ends) < 3:
    sys.exit("no steps found")
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t_prev = int(rows[lo - 1]["End_Timestamp"])
tot_k = tot_g = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    print("%-72s %8.1f us  gap %6.1f  grid %s wg %s" % (n, (e - s) / 1e3, (s - t_prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    tot_k += e - s; tot_g += max(0, s - t_prev); t_prev = e
print("kernels %d  busy %.1f us  gaps %.1f us  span %.1f us" % (len(step), tot_k / 1e3, tot_g / 1e3, (int(step[-1]["End_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])) / 1e3))
