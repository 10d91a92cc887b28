"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/ops_only.py -> profiles/r03_ops_pmc.json: memory-side bytes per launch of every
op of bench.py's roofline_ops leg.  Kernels between a marker (three_nn_weights_kernel, grid 256*(k+101)) and the end marker (grid 256*100)
belong to key k.  FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request: MI355X_MICROARCH.md, HBM section); both are in KiB."""
import csv, glob, json, sys
fetch_dir, write_dir, log, out_path = sys.argv[1:5]
keys = json.loads([l for l in open(log) if l.startswith("KEYS ")][-1][5:])
res = {k: {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "kernels": []} for k in keys}
for d, name in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    cur = None
    for r in rows:
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if kn.startswith("three_nn_weights_kernel"):
            k = int(r["Grid_Size"]) // 256 - 101
            cur = keys[k] if 0 <= k < len(keys) else None
            continue
        if cur is None:
            continue
        res[cur][name + "_KiB"] += float(r["Counter_Value"])
        if name == "FETCH_SIZE":
            res[cur]["kernels"].append(kn[:60])
ops = {k: (2.0 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024.0 for k, v in res.items()}
json.dump({"ops": ops, "detail": res, "notes": "memory-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024, separate --pmc passes; "
           "kernels = what ran inside the bracket"}, open(out_path, "w"), indent=1)
for k in keys:
    print("%-32s %12.0f B  %s" % (k, ops[k], ",".join(sorted(set(res[k]["kernels"])))[:100]))
