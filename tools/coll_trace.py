"""which hardware queue did each kernel of a traced bench run use?  reads a rocprofv3 kernel_trace.csv; prints, per Queue_Id, the kernel
families seen there and their counts (diagnosis of the +1.1 ms per step that the forced collective cost in r03)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))][k] += 1
for q in sorted(byq):
    tot = sum(byq[q].values())
    print("queue %s stream %s: %d kernels: %s" % (q[0], q[1], tot, ", ".join("%s x%d" % kv for kv in byq[q].most_common(6))))
