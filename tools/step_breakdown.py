"""Steady-state per-kernel breakdown of one training step from a rocprofv3 kernel trace (steps delimited by sgd_kernel)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"] or "sgd_dev_kernel" in r["Kernel_Name"]]
if len(idx) < 3:
    raise SystemExit("need >= 3 optimizer steps in the trace")
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = r["Kernel_Name"][:int(sys.argv[3]) if len(sys.argv) > 3 else 100]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"step span {span:.2f} ms, busy {sum(v[1] for v in agg.values()):.2f} ms, {len(step)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{v[1]:8.3f} ms {v[0]:5d}  {k}")
