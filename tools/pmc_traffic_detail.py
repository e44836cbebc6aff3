"""Per-kernel means of the counters collected by tools/pmc_traffic_detail.sh (one line per kernel and counter group)."""
import collections, csv, glob, re, sys

base = sys.argv[1]


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:60]


vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for d in sorted(glob.glob(base + "/p*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(vals, key=lambda k: -sum(durs[k])):
    if not any(t in k for t in ("modconv", "blur", "to_rgb", "wino")):
        continue
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    n = len(next(iter(vals[k].values())))
    us = sum(durs[k]) / len(durs[k]) / 1e3
    rd = c.get("TCC_EA0_RDREQ_sum", 0)
    r32, r64, r128 = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
    other = rd - r32 - r64 - r128
    rd_bytes = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(other, 0)
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print(f"{k}: launches {n}, avg {us:.1f} us")
    print(f"    FETCH_SIZE {c.get('FETCH_SIZE', 0) * 1024 / 1e6:10.1f} MB (raw, x1024 B)   WRITE_SIZE {c.get('WRITE_SIZE', 0) * 1024 / 1e6:10.1f} MB")
    print(f"    RDREQ {rd:12.0f} = 32B {r32:12.0f} + 64B {r64:12.0f} + 128B {r128:12.0f} + unsized {other:12.0f} -> {rd_bytes / 1e6:10.1f} MB read at the L2's memory side")
    print(f"    RDREQ_DRAM {c.get('TCC_EA0_RDREQ_DRAM_sum', 0):12.0f}  RD_UNCACHED_32B {c.get('TCC_EA0_RD_UNCACHED_32B_sum', 0):10.0f}  "
          f"WRREQ {c.get('TCC_EA0_WRREQ_sum', 0):12.0f} (64B {c.get('TCC_EA0_WRREQ_64B_sum', 0):12.0f}, DRAM {c.get('TCC_EA0_WRREQ_DRAM_sum', 0):12.0f})")
    print(f"    TCC_HIT {hit:14.0f}  TCC_MISS {miss:14.0f}  hit rate {hit / max(hit + miss, 1):.3f}")
