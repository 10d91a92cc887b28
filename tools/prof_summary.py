"""summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-step totals by kernel"""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms/step over %g steps" % (tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-64s calls/step %6.1f avg %8.1f us  %6.3f ms/step %5.1f%%" % (r['Name'][:64], float(r['Calls']) / steps, float(r['AverageNs']) / 1e3,
          float(r['TotalDurationNs']) / 1e6 / steps, float(r['Percentage'])))
