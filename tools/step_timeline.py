"""Development tool: gaps and overlaps of one synthesis step from a rocprofv3 kernel trace CSV.
usage: python tools/step_timeline.py <kernel_trace.csv> [launches per step = auto]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# steps: split on the first kernel of the generator (pixel_norm) -- take the last full step
starts = [i for i, n in enumerate(names) if "pixel_norm" in n]
if len(starts) < 3:
    sys.exit("need at least 3 steps in the trace")
# (a step of the timed region: of all consecutive pairs of step starts, the one with the median span -- the trace also holds
# warm-up, parity and CPU-baseline stages with the same first kernel)
pairs = sorted(zip(starts[:-1], starts[1:]), key=lambda ab: int(rows[ab[1]]["Start_Timestamp"]) - int(rows[ab[0]]["Start_Timestamp"]))
a, b = pairs[len(pairs) // 2]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"step: {len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms")
by_stream = defaultdict(list)
for r in step:
    by_stream[r.get("Stream_Id", r.get("Queue_Id", "0"))].append(r)
for sid, rs in by_stream.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"stream/queue {sid}: {len(rs)} launches, busy {busy / 1e6:.3f} ms")
# union busy time and idle gaps
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cur_s, cur_e, busy, gaps = iv[0][0], iv[0][1], 0, []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (union) {busy / 1e6:.3f} ms, idle {sum(g for g, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; largest gaps (us @ ms):",
      [(round(g / 1e3, 1), round(at / 1e6, 2)) for g, at in sorted(gaps, reverse=True)[:8]])
print("launches in order (ms from step start, duration us, name):")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  {(s - t0) / 1e6:7.3f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
