"""per kernel of a rocprofv3 --pmc run (SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR): calls, average
microseconds, instruction counts per launch and the SIMD floor they imply on gfx950 -- fp32 MFMAs (64 cycles) and vector instructions (~2.8, the v_fma rate of tools/clk/coexec.hip)
of all waves ADD on a SIMD (tools/clk/coexec.hip) -- as microseconds over 1024 SIMDs at 2.3 GHz, and its share of the measured time.
usage: simd_budget.py <rocprof dir> [min_us]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[k]) / len(dur[k])
    floor = (64.0 * m.get("SQ_INSTS_MFMA", 0) + 2.8 * m.get("SQ_INSTS_VALU", 0)) / 1024 / 2300.0
    rows.append((us * len(dur[k]), k, len(dur[k]), us, m, floor))
rows.sort(reverse=True)
print("%-62s %5s %8s %9s %9s %9s %7s %8s %6s" % ("kernel", "calls", "us", "MFMA", "VALU", "LDS", "V/M", "simd_us", "share"))
for tot, k, n, us, m, floor in rows:
    if us < min_us:
        continue
    mf = m.get("SQ_INSTS_MFMA", 0)
    print("%-62s %5d %8.1f %9.0f %9.0f %9.0f %7.1f %8.1f %6.2f" % (k[:62], n, us, mf, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_LDS", 0),
                                                                  m.get("SQ_INSTS_VALU", 0) / mf if mf else 0.0, floor, floor / us))
