"""Per-kernel HBM traffic from the PMC passes of tools/pmc_traffic.sh.

FETCH_SIZE / WRITE_SIZE are in units of 1 KB as rocprofv3 reports them; on gfx950 FETCH_SIZE tallies each 128-byte request
of a wide coalesced read at 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM").  Output: JSON
{"kernels": {name: {launches, fetch_bytes, write_bytes, traffic_bytes (all per launch, mean), workload}}} where ``name`` is
the label bench.py uses for the kernel (the rocprofv3 name with the template wrapper types folded away), so that
``roofline.traffic`` of every workload can be looked up by its dominant kernel.
"""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "synthesis-in-style_amd"))
base, out, workloads = sys.argv[1], sys.argv[2], (sys.argv[3:] or ["synthesis"])


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:120]


def label(name):
    """rocprofv3 kernel name -> the label of the same launches in bench.py's per-kernel records."""
    m = re.match(r"gemm_bf16_kernel<GemmCfg<(\d+), (\d+), (true|false), (true|false)(?:, \d+, \d+)?\s*>, (\d+)>", name)
    if m:
        layout = {("false", "false"): "NT", ("false", "true"): "NN", ("true", "true"): "TN"}.get((m.group(3), m.group(4)), "??")
        return f"gemm_bf16<{layout},{m.group(5)}>"
    m = re.match(r"gemm256_kernel<G256Cfg<(\d+)>, (\d+)>", name)
    if m:
        return f"gemm256<{96 * int(m.group(1))},{m.group(2)}>"
    m = re.match(r"conv_bf16_kernel<ConvCfg<([^>]*)>", name)
    if m:
        return "conv_bf16_kernel<" + m.group(1).replace(" ", "") + ">"
    m = re.match(r"conv1x1_f32_kernel<PwCfg<([^>]*)>", name)
    if m:
        return "conv1x1_f32_kernel<" + m.group(1).replace(" ", "") + ">"
    m = re.match(r"conv_wgrad_bf16_kernel<WgCfg<([^>]*)>", name)
    if m:
        return "conv_wgrad_bf16_kernel<" + m.group(1).replace(" ", "") + ">"
    m = re.match(r"(modconv_wino2_kernel|conv1x1_wgrad_bf16_kernel|conv1x1_wgrad_f32_kernel|attn_fwd_kernel|emau_kernel)\b", name)
    if m:
        return m.group(1)
    return name


def per_kernel(sub, counter):
    f = glob.glob(f"{sub}/*/*counter_collection.csv") + glob.glob(f"{sub}/*counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                acc[label(short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
    return acc


import sis_hip  # noqa: E402  (own_kernel_names only: no device needed)

res_by_workload = {}
for w in workloads:
    res = res_by_workload.setdefault(w, {})
    root = f"{base}/{w}" if os.path.isdir(f"{base}/{w}") else base
    fetch, write = per_kernel(f"{root}/fetch", "FETCH_SIZE"), per_kernel(f"{root}/write", "WRITE_SIZE")
    for k in sorted(set(fetch) | set(write)):
        if not sis_hip.is_own_kernel(k.replace("gemm_bf16<", "gemm_bf16_kernel<").replace("gemm256<", "gemm256_kernel<")):
            continue
        fb = 2.0 * 1024.0 * (sum(fetch[k]) / len(fetch[k])) if fetch.get(k) else None
        wb = 1024.0 * (sum(write[k]) / len(write[k])) if write.get(k) else None
        entry = {"launches": len(fetch.get(k) or write.get(k)), "fetch_bytes": fb, "write_bytes": wb,
                 "traffic_bytes": (fb or 0.0) + (wb or 0.0), "workload": w}
        res[k] = entry
# "kernels": the synthesis workload (bench.py's headline roofline); "training": {workload: {kernel: ...}}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --workload W --steps 2 --warmup 1` "
                     "(training steps eager); FETCH_SIZE x2 (gfx950), x1024 B", "kernels": res_by_workload.get("synthesis", {}),
           "training": {w: r for w, r in res_by_workload.items() if w != "synthesis"}}, open(out, "w"), indent=1)
for w, res in res_by_workload.items():
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:40]:
        print(f"{w:10s} {k:52s} n={v['launches']:4d} fetch {(v['fetch_bytes'] or 0)/1e6:9.1f} MB  write {(v['write_bytes'] or 0)/1e6:9.1f} MB")
