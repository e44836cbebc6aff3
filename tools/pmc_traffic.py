"""Per-kernel HBM traffic from the two PMC passes of tools/pmc_traffic.sh.

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB as rocprofv3 reports them; on gfx950 FETCH_SIZE tallies each
128-byte request of a wide coalesced read at 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM").  Output: JSON
{kernel short name: {launches, fetch_bytes, write_bytes, traffic_bytes (all per launch, mean)}}.
"""
import collections, csv, glob, json, re, sys

base, out = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def per_kernel(sub, counter):
    f = glob.glob(f"{base}/{sub}/*/*counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ("modconv", "blur", "to_rgb", "wino")):
        continue
    fb = 2.0 * 1024.0 * (sum(fetch[k]) / len(fetch[k])) if fetch.get(k) else None
    wb = 1024.0 * (sum(write[k]) / len(write[k])) if write.get(k) else None
    res[k] = {"launches": len(fetch.get(k) or write.get(k)), "fetch_bytes": fb, "write_bytes": wb,
              "traffic_bytes": (fb or 0.0) + (wb or 0.0)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1`; "
                     "FETCH_SIZE x2 (gfx950), x1024 B", "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:40s} n={v['launches']:4d} fetch {(v['fetch_bytes'] or 0)/1e6:9.1f} MB  write {(v['write_bytes'] or 0)/1e6:9.1f} MB")
