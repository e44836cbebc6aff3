"""Summarise rocprofv3 --pmc CSVs produced by tools/pmc_layers.sh: per kernel, mean counter values per dispatch."""
import csv, glob, sys, collections
base = sys.argv[1]
for d in sorted(glob.glob(base + "/p*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    t = glob.glob(d + "*/*kernel_trace.csv")
    if not f:
        continue
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t[0])):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "modconv_" not in k and len(sys.argv) < 3:
            continue
        ds = dur.get(k, [0])
        print(f"{d} {k[:70]} n={len(ds)} avg_us={sum(ds)/len(ds)/1e3:.1f}")
        for c, v in cs.items():
            print(f"    {c:32s} {sum(v)/len(v):.4g}")
