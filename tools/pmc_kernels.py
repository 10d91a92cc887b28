"""SQ counter ratios per kernel from a rocprofv3 --pmc run (WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY = issue stall)"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
pat = sys.argv[2:] or [""]                     # no pattern: every kernel (r04: the r03 filter missed the lean / fused kernels)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if any(p in k for p in pat):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    busy = m.get("SQ_BUSY_CYCLES", 1) / 32.0 * 1024      # SIMD-cycles of the launch (32 SEs report busy cycles)
    print("%-50s x%-4d %6.1f us  " % (k.replace("void ", "")[:50], len(set(dur[k])) and len(dur[k]) // max(1, len(c)), sum(dur[k]) / len(dur[k])) + " ".join("%s=%.2f" % (n.replace("SQ_", ""), m[n] / wc) for n in sorted(m) if n not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"))
          + "  mfma_util=%.2f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy))
