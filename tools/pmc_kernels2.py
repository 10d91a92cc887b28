"""instruction-mix counters per kernel from a rocprofv3 --pmc run: VALU / MFMA / LDS instructions per wave, LDS bank-conflict share"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
pat = sys.argv[2:] or ["wgrad_stream", "mlp_fwd", "mlp_bwd_data"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if any(p in k for p in pat):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    mf = max(m.get("SQ_INSTS_MFMA", 0), 1)
    print("%-50s %7.1f us  VALU/MFMA=%.2f LDS/MFMA=%.2f  lds_bank_conflict=%.2f of LDS-active cycles" %
          (k[-50:], sum(dur[k]) / len(dur[k]), m.get("SQ_INSTS_VALU", 0) / mf, m.get("SQ_INSTS_LDS", 0) / mf,
           m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
