"""Every launch of one steady-state training step, in order, from a rocprofv3 kernel trace (steps delimited by the SGD kernel):
start (ms from the step's start), duration (us), grid / workgroup size, kernel name -- for mapping time to LAYERS.
usage: step_launches.py <rocprof dir> [name width]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 110
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"] or "sgd_dev_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
print(f"{len(step)} launches, span {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:8.1f} {r.get('Grid_Size_X', '?'):>9s}x{r.get('Grid_Size_Y', '?'):<5s} {r.get('Workgroup_Size_X', '?'):>5s}  {name[:width]}")
