"""Steady-state per-kernel breakdown of the GAN-training bench from a rocprofv3 kernel trace: only the launches inside
the timed region (the last steps * ms_per_step of the trace -- everything before is warm-up and the library's solver
search).  usage: python tools/gan_breakdown.py <rocprof dir> <bench json> [top]"""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
span_ns = bench["ms_per_step"] * bench["steps"] * 1e6
rows = list(csv.DictReader(open(f)))
end = max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r["Start_Timestamp"]) >= end - span_ns:
        k = r["Kernel_Name"][:110]
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
busy = sum(v[1] for v in agg.values())
print(f"timed region {span_ns / 1e6:.1f} ms ({bench['steps']} iterations), kernel busy {busy:.1f} ms, "
      f"{sum(v[0] for v in agg.values())} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{v[1]:9.2f} ms {100 * v[1] / busy:5.1f}% {v[0]:6d}  {k}")
