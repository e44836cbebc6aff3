"""Headline benchmark: StyleGAN2 256x256 synthesis (Generator.forward) images/s on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches one
rank per GPU through torch.distributed.run.  W untimed warm-up steps, then exactly K timed steps
bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks, ONE JSON line from
rank 0.

Workload (BASELINE.json configs[1]): Generator(256, 512, 8, channel_multiplier=2).forward on a batch of
32 latents per GPU, explicit noise maps, ``return_intermediate_activations=True`` (the mode
create_dataset_for_segmentation.py runs, utils/dataset_creation.py:50-57), fp32, synthetic seeded
weights (default init, noise weights ~N(0, 0.1^2) so the noise path is live), inputs resident in HBM.
A "step" is one such forward.  Synthesis shards by image with no collective, so N GPUs
= N independent batches per step: weak scaling; value = N * 32 * K / time.

With the default ``--workload all`` (what the driver runs) the same process then times the two segmentation training
steps the metric also names -- BASELINE.json configs[3] EMANet-50 256^2 B=16 fp32 and configs[4] TransUNet R50-ViT-B/16
512^2 B=8 bf16 -- with the same K, and nests them as ``seg_train: {emanet: {...}, transunet_bf16: {...}}`` (images/s,
ms/step, roofline with executed vs nominal FLOPs, cpu_baseline); ``value`` stays the synthesis rate.

Two extra objects on the JSON line:
  roofline      dominant kernel (by summed device time) measured live with HIP events on the launch
                stream: EXECUTED FLOPs per launch / average launch duration vs the fp32 MFMA peak (frac <= 1); the
                direct-form (algorithmic) count of SURVEY.md 8(d) rides along as algorithmic_tflops / algorithmic_frac.
  cpu_baseline  the oracle (CPU restatement of the reference's own grouped-conv formulation,
                oracle/stylegan2_ref.py) timed on the host cores of this box on a bounded sample
                (batch 4), rank 0 / N = 1 only.  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "synthesis-in-style_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "StyleGAN2 synth images/sec + seg-train images/sec @256², 1/2/4/8 MI355X"
PEAK_MFMA_F32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_MFMA_BF16_TFLOPS = 2500.0  # dense, same table
PEAK_HBM_GBS = 8000.0
BATCH = 32
SIZE = 256


def build_generator(device, seed=0):
    from networks.stylegan2.model import Generator
    torch.manual_seed(seed)
    g = Generator(SIZE, 512, 8, channel_multiplier=2)
    with torch.no_grad():
        for name, p in g.named_parameters():
            if name.endswith("noise.weight"):
                p.normal_(0.0, 0.1)
    return g.to(device).eval()


def synth_inputs(g, batch, device, seed):
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(batch, 512, generator=gen).to(device)
    sizes = [4] + [2 ** i for i in range(3, g.log_size + 1) for _ in range(2)]
    noise = [torch.randn(1, 1, s, s, generator=gen).to(device) for s in sizes]
    return z, noise


def step(g, z, noise):
    with torch.no_grad():
        return g([z], noise=noise, return_intermediate_activations=True)


def exec_ratio(kernel_name):
    """Executed / direct-form matrix FLOPs of a kernel: Winograd F(2x2,3x3) runs 16 of the 36 multiplies of a 3x3 convolution,
    the fast-FIR transposed convolution 25 of 36 (csrc/modconv_upfir.hip); launches are recorded in the direct-form count of
    SURVEY.md 8(d)."""
    if "wino" in kernel_name:
        return 16.0 / 36.0
    if "upfir" in kernel_name:
        return 25.0 / 36.0
    return 1.0


def kernel_profile(g, z, noise, steps):
    """Per-kernel device time from HIP events recorded on the launch stream around every launch."""
    import sis_hip
    records = []
    sis_hip.set_profiler(records)
    try:
        for _ in range(steps):
            step(g, z, noise)
        torch.cuda.synchronize()
    finally:
        sis_hip.set_profiler(None)
    agg = {}
    for name, flops, nbytes, e0, e1 in records:
        a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        a["launches"] += 1
        a["ms"] += e0.elapsed_time(e1)
        a["flops"] += flops
        a["bytes"] += nbytes
    return agg


def host_cores():
    """CPU share of this process: min(affinity mask, cgroup quota); the GPU box exposes 256 logical CPUs
    but grants a 16-core share per GPU, and oversubscribing torch's intra-op pool makes the baseline
    slower, not faster."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("SIS_BENCH_CPU_THREADS", "16")))


def cpu_baseline(g, seed):
    """Oracle on the host cores: B=4 (BASELINE.json configs[0]), 1 warm-up + 3 timed iterations, median."""
    from oracle import stylegan2_ref as R
    sd = {k: v.detach().cpu() for k, v in g.state_dict().items()}
    z, noise = synth_inputs(g, 4, "cpu", seed)
    threads = host_cores()
    torch.set_num_threads(threads)
    with torch.no_grad():
        img, _ = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            img, _ = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
            t.append(time.perf_counter() - t0)
    sec = sorted(t)[1]
    return {"value": round(4 / sec, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"Generator(256,512,8,cm=2) forward with activations, batch 4, fp32, {threads} torch threads, "
                      f"median of 3 after 1 warm-up ({sec:.2f} s/iter)"}, img, (z, noise)


SEG_FLOPS_PER_IMAGE = {"emanet": 227.38e9, "transunet": 1007.79e9}  # fwd+bwd, SURVEY.md §8(d) (2*MAC)
SEG_CONFIG = {"emanet": "configs/segmenter/ema_net_resnet50_256.yaml",
              "transunet": "configs/segmenter/trans_u_net_r50_vit_b16_512.yaml"}


def cpu_baseline_training(workload, config):
    """Oracle training step on the host cores (B = 4 for EMANet-50 @256^2, B = 2 for TransUNet @512^2,
    fp32): 1 warm-up + 3 timed iterations (median), bounded to tens of seconds."""
    threads = host_cores()
    torch.set_num_threads(threads)
    size, classes = config["image_size"], config["num_classes"]
    if workload == "emanet":
        from oracle import ema_net_ref as O
        batch_size, sd = 4, O.seeded_state_dict(50, classes, seed=0)
        step = lambda b: O.train_step(sd, bufs, b)  # noqa: E731
    else:
        from oracle import trans_u_net_ref as O
        batch_size, sd = 2, O.seeded_state_dict(size, classes, seed=0)
        step = lambda b: O.train_step(sd, bufs, b, num_classes=classes)  # noqa: E731
    gen = torch.Generator().manual_seed(7)
    batch = {"images": torch.rand(batch_size, 3, size, size, generator=gen) * 2 - 1,
             "segmented": torch.randint(0, classes, (batch_size, 1, size, size), generator=gen)}
    bufs = {}
    step(batch)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        step(batch)
        times.append(time.perf_counter() - t0)
    sec = sorted(times)[1]
    return {"value": round(batch_size / sec, 3), "unit": "images/s", "cores": threads, "kind": "port", "batch": batch_size,
            "sample": f"{config['network']} oracle training step (forward, loss, backward, SGD), {size}x{size}, batch "
                      f"{batch_size}, fp32, {threads} torch threads, median of 3 iterations after 1 warm-up ({sec:.2f} s)"}


def _graph_expected(updater):
    """The updater was built to replay its iteration as a hipGraph (single process, or a capturable gradient exchange)."""
    graph = getattr(updater, "_step_graph", None)
    return graph is not None and graph.requested


def library_time(updater):
    """Device time of ONE training iteration split into this library's kernels and everything else (ATen / MIOpen / hipBLASLt /
    runtime copies), from the kernel records of torch.profiler (roctracer) around one more EAGER ``update()`` (the same
    kernels a graph replay runs; the tracer does not see inside a replay).  A kernel counts as own when its name contains a ``__global__`` function of csrc/*.hip
    (sis_hip.own_kernel_names).  None when the profiler delivers no device records on this box."""
    import sis_hip
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return {"skipped": "running under rocprofv3 (two tracers in one process); see profiles/ for the external breakdown"}
    graph = getattr(updater, "_step_graph", None)
    was_enabled = graph.enabled if graph is not None else None
    try:
        from torch.profiler import ProfilerActivity, profile
        if graph is not None:
            graph.enabled = False   # one EAGER iteration: the tracer does not see inside a graph replay; the kernels are the same
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            updater.update()
            torch.cuda.synchronize()
        if graph is not None:
            graph.enabled = was_enabled
        own_us = lib_us = 0.0
        lib_names = {}
        n = 0
        for ev in prof.events():
            if getattr(ev, "device_type", None) is None or "cuda" not in str(ev.device_type).lower():
                continue
            if getattr(ev, "is_user_annotation", False) or "#" in ev.name:   # annotation ranges mirrored onto the device timeline
                continue
            dur = float(getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0) or 0.0)
            if dur <= 0.0:
                continue
            n += 1
            if sis_hip.is_own_kernel(ev.name):
                own_us += dur
            else:
                lib_us += dur
                lib_names[ev.name[:60]] = lib_names.get(ev.name[:60], 0.0) + dur
        if n == 0:
            return None
        top = sorted(lib_names.items(), key=lambda kv: -kv[1])[:6]
        return {"own_ms": round(own_us / 1e3, 3), "library_ms": round(lib_us / 1e3, 3), "device_records": n,
                "top_library_kernels_ms": {k: round(v / 1e3, 3) for k, v in top}}
    except Exception as err:   # the measurement must never take the bench line down
        if graph is not None and was_enabled is not None:
            graph.enabled = was_enabled
        return {"error": repr(err)}


class single_rank_communicator:
    """N = 1 only: a world-size-1 RCCL communicator for the duration of the data-parallel rehearsal, and not a moment longer --
    the bare steps (and the synthesis rate, which is timed first) are measured without it."""

    def __init__(self, device):
        self.device, self.created = device, False

    def __enter__(self):
        import socket
        import torch.distributed as dist
        if dist.is_initialized():
            return self
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=self.device)
            self.created = True
        except Exception as err:   # no RCCL on this box: the rehearsal is skipped, the N = 1 measurements do not need it
            print(f"bench.py: RCCL world-size-1 communicator unavailable ({err!r}); data-parallel rehearsal skipped", file=sys.stderr)
        return self

    def __exit__(self, *exc):
        import torch.distributed as dist
        if self.created and dist.is_initialized():
            import gc
            gc.collect()   # (the rehearsal's captured step graphs hold the communicator's collectives: gone before it goes)
            torch.cuda.synchronize()
            dist.destroy_process_group()
        return False


def data_parallel_rehearsal(args, workload, config, device):
    """What the N > 1 per-GPU step looks like, as far as ONE GPU can show it: the same training step with the network inside
    the data-parallel wrap over a world-size-1 RCCL communicator (bucketed reduce-scatter + all-gather on RCCL's stream,
    FusedSGD on the bucket views), once replayed as a hipGraph (collectives captured) and once eager.  No scaling claim: one
    rank exchanges nothing; it shows that the path runs on RCCL and what the wrap costs per step."""
    import gc
    import torch.distributed as dist
    from training_builder.train_builder_selection import get_train_builder_class
    from utils.synthetic_data import SyntheticSegmentationLoader
    if not dist.is_initialized():
        return None
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "flavour": config.get("data_parallel", "buckets")}
    import training.grad_exchange as grad_exchange
    saved_direct = grad_exchange._DIRECT_RCCL
    # "graph" / "eager": collectives issued straight into librccl.so (the default at world size 1; captured = "graph");
    # "torch_collectives_eager": the same buckets through torch.distributed's work objects (the default at world size > 1)
    only = os.environ.get("SIS_BENCH_DP_MODES")   # profiling: a comma-separated subset of the legs
    for mode, hip_graph, direct in (("graph", True, "1"), ("eager", False, "1"), ("torch_collectives_eager", False, "0")):
        if only and mode not in only.split(","):
            continue
        grad_exchange._DIRECT_RCCL = direct
        gc.collect()
        torch.cuda.empty_cache()
        cfg = dict(config, force_data_parallel=True, hip_graph=hip_graph)
        loader = SyntheticSegmentationLoader(cfg["batch_size"], cfg["image_size"], cfg["num_classes"], seed=1234, device=device)
        torch.manual_seed(0)
        builder = get_train_builder_class(cfg)(cfg, loader, None, rank=device.index, world_size=1)
        updater = builder.get_updater()
        if hip_graph:
            updater._step_graph.strict = True
        for _ in range(max(args.warmup, 4)):
            updater.update()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            updater.update()
        torch.cuda.synchronize()
        out[f"{mode}_ms_per_step"] = round((time.perf_counter() - t0) / args.steps * 1e3, 3)
        net = builder.get_network()
        out["buckets"] = len(net.buckets or [])
        out["collective"] = net.collective
        if hip_graph:
            out["hip_graph"] = updater._step_graph.graph is not None
        if direct == "1":
            out["direct_rccl"] = net.direct_rccl()
            out["direct_rccl_note"] = net.direct_rccl_note
            moved = net.stats["copied_elems"] + net.stats["in_place_elems"]
            # share of the gradient elements the weight-gradient kernels wrote straight into the buckets (eager backwards
            # after the discovery one; replays run the same launches)
            per = moved / max(net.stats["backwards"], 1)
            out["gradients_written_in_place"] = round(net.stats["in_place_elems"] / max(moved - per, 1), 4)
        del updater, builder, net
    grad_exchange._DIRECT_RCCL = saved_direct
    out["default_at_world_size_gt_1"] = "eager, torch.distributed collectives (SIS_DP_DIRECT_RCCL=1: direct RCCL calls behind a self-check)"
    return out


def bench_training(args, workload, world, rank, device, distributed):
    """Segmentation training images/s (BASELINE.json configs[3] / [4]): one step = one updater iteration
    (forward, loss, backward with bucketed RCCL all-reduce, fused SGD step) on a synthetic batch resident in HBM."""
    import yaml
    import torch.distributed as dist
    from training_builder.train_builder_selection import get_train_builder_class
    from utils.synthetic_data import SyntheticSegmentationLoader
    config = yaml.safe_load(open(os.path.join(ROOT, "synthesis-in-style_amd", SEG_CONFIG[workload])))
    config["fine_tune"] = None
    if args.dtype:
        if workload == "emanet" and args.dtype != "f32":
            raise SystemExit("the EMANet row is fp32, the reference's precision (BASELINE.json configs[3]); only the TransUNet "
                             "row (configs[4]) names bf16")
        config["amp"] = None if args.dtype == "f32" else args.dtype
    if args.batch:
        config["batch_size"] = args.batch
    if args.miopen_search:
        config["miopen_search"] = True
    loader = SyntheticSegmentationLoader(config["batch_size"], config["image_size"], config["num_classes"],
                                         seed=1234 + rank, device=device)
    torch.manual_seed(0)
    builder = get_train_builder_class(config)(config, loader, None, rank=device.index, world_size=world)
    updater = builder.get_updater()
    graph = getattr(updater, "_step_graph", None)
    if graph is not None and world == 1:
        graph.strict = True   # a capture that fails must fail the bench, not pass as an eager measurement

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Iterations 1-2 are eager (StepGraph warm-up); the second one is bracketed kernel by kernel with HIP events on the
    # launch stream: nominal FLOPs and device time of every hand-written convolution launch of one step.
    import sis_hip
    warmup = max(args.warmup, 4)  # two eager iterations, the capture, one replay before the clock starts
    updater.update()
    records = []
    sis_hip.library_calls(reset=True)
    sis_hip.set_profiler(records)
    try:
        updater.update()
        torch.cuda.synchronize()
    finally:
        sis_hip.set_profiler(None)
    library_calls = sis_hip.library_calls(reset=True)   # one whole iteration: what was handed to ATen / MIOpen / hipBLASLt
    own = {}
    for name, flops, nbytes, e0, e1 in records:
        a = own.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        a["launches"] += 1
        a["ms"] += e0.elapsed_time(e1)
        a["flops"] += flops
        a["bytes"] += nbytes
    for _ in range(warmup - 2):
        updater.update()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        updater.update()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    hip_graph = bool(graph is not None and graph.graph is not None)
    if graph is not None and graph.requested and not hip_graph and world == 1 and _graph_expected(updater):
        raise SystemExit(f"bench.py: the {workload} step was to be replayed as a hipGraph and was not captured "
                         f"({graph.capture_error}); refusing to report an eager measurement as the graphed one")
    baseline_config = not args.batch and ((workload == "emanet" and not config.get("amp")) or
                                          (workload == "transunet" and config.get("amp") == "bf16"))
    if baseline_config and library_calls["fallback"]:
        raise SystemExit(f"bench.py: {workload} handed operators to the ROCm libraries that the BASELINE config is supposed to "
                         f"run on libsis_hip.so: {library_calls['fallback']}")
    # (one more iteration on rank 0 alone: only where an iteration holds no collective, i.e. in a single process)
    library = library_time(updater) if (rank == 0 and world == 1) else None
    dp = None
    if world == 1 and rank == 0 and args.dp_rehearsal:
        del updater, builder
        with single_rank_communicator(device):
            dp = data_parallel_rehearsal(args, workload, config, device)
    if rank != 0:
        return None
    if dp is not None and baseline_config and dp.get("backend") == "nccl" and dp.get("direct_rccl") is False:
        # the rehearsal's graph leg was to run on the direct librccl.so path: a torch build without _comm_ptr (or no mapped
        # RCCL) silently measuring the work-object path would pass as the captured exchange
        raise SystemExit(f"bench.py: {workload}: the data-parallel rehearsal did not reach librccl.so directly "
                         f"(direct_rccl: false, {dp.get('direct_rccl_note')})")
    images = config["batch_size"] * args.steps * world
    step_flops = SEG_FLOPS_PER_IMAGE[workload] * config["batch_size"]           # nominal 2*MAC, SURVEY.md §8(d)
    # EXECUTED matrix FLOPs of the hand-written kernels, as recorded at their launches in the instrumented eager iteration:
    # Winograd F(2x2,3x3) launches execute 16 of the 36 multiplies of the direct form they are recorded in; stride-2 layers run
    # as dense stride-1 launches and dilated ones on sub-images are recorded (and counted) with the work they really do; the
    # attention backward is recorded with its 7 products (2 recomputed).  What still runs on vendor libraries (a 3-channel stem
    # convolution, a few small GEMMs) is NOT in this sum: the number is a lower bound of the executed rate.
    executed_flops = sum(v["flops"] * exec_ratio(k) for k, v in own.items())
    step_s = elapsed / args.steps
    peak = PEAK_MFMA_BF16_TFLOPS if config.get("amp") else PEAK_MFMA_F32_TFLOPS
    nominal_tf, executed_tf = step_flops / step_s / 1e12, executed_flops / step_s / 1e12
    matrix = {k: v for k, v in own.items() if v["flops"]}
    dom = max(matrix, key=lambda k: matrix[k]["ms"]) if matrix else None
    traffic, traffic_src = measured_traffic(dom, workload) if dom else (None, None)

    def kernel_peak(name):   # fp32 kernels (Winograd, fp32 1x1) inside an AMP step are priced against the fp32 peak
        return PEAK_MFMA_F32_TFLOPS if ("wino" in name or "f32" in name or not config.get("amp")) else PEAK_MFMA_BF16_TFLOPS

    def kernel_row(name, v):
        if not v["ms"]:
            return {"launches": v["launches"], "ms": 0.0}
        sec = v["ms"] * 1e-3
        executed = v["flops"] * exec_ratio(name)
        tf, gbs = executed / sec / 1e12, v["bytes"] / sec / 1e9
        ridge = kernel_peak(name) * 1e12 / (PEAK_HBM_GBS * 1e9)      # FLOP per byte at which the two roofs meet
        hbm_bound = v["bytes"] > 0 and (v["flops"] == 0 or executed / v["bytes"] < ridge)
        return {"launches": v["launches"], "ms": round(v["ms"], 3),
                "nominal_tflops": round(v["flops"] / sec / 1e12, 1), "gbs_algorithmic": round(gbs, 1),
                "bound": "hbm" if hbm_bound else "mfma",
                "frac_of_peak": round(gbs / PEAK_HBM_GBS, 4) if hbm_bound else (round(tf / kernel_peak(name), 4) if v["flops"] else None)}
    return {
        "metric": METRIC, "value": round(images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": config.get("amp") or "f32", "data": "synthetic",
        "config": {"workload": f"{config['network']} training step, {config['image_size']}x{config['image_size']}, batch "
                               f"{config['batch_size']} per GPU (BASELINE.json configs[{3 if workload == 'emanet' else 4}])",
                   "batch_per_gpu": config["batch_size"], "image_size": config["image_size"],
                   "parallelism": f"dp{world}, bucketed reduce-scatter + all-gather over RCCL (training/grad_exchange.py)",
                   "hip_graph": hip_graph, "graph_capture_error": None if graph is None else graph.capture_error,
                   "miopen_search": bool(config.get("miopen_search"))},
        "library_calls_per_step": library_calls, "library_ms_per_step": library,
        "data_parallel_rehearsal": dp,
        "roofline": {"kernel": "whole training step (forward, loss, backward, SGD)", "bound": "mfma",
                     "achieved": round(executed_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(executed_tf / peak, 4),
                     "note": "achieved / frac = EXECUTED matrix FLOPs of the hand-written kernels (recorded per launch; Winograd "
                             "launches at 16/36 of their direct-form count; library remainder excluded) / step time; "
                             "algorithmic_* use the nominal 2*MAC count of SURVEY.md 8(d)",
                     "algorithmic_tflops": round(nominal_tf, 2), "algorithmic_frac": round(nominal_tf / peak, 4),
                     "flops_per_step_nominal": step_flops, "flops_per_step_executed": executed_flops,
                     "traffic": traffic, "traffic_unit": "HBM-side bytes per launch of dominant_own_kernel (PMC FETCH_SIZE x 2 + "
                                                        "WRITE_SIZE, separate passes)", "traffic_source": traffic_src,
                     "dominant_own_kernel": dom,
                     # per kernel: the roofline that bounds it -- "hbm" when its algorithmic FLOP / byte sits below the ridge of
                     # its matrix peak (1x1 convolutions on wide maps, norms), else "mfma" -- and its fraction of THAT peak
                     "own_kernels_eager_iteration": {k: kernel_row(k, v) for k, v in sorted(own.items(), key=lambda kv: -kv[1]["ms"])}},
        "cpu_baseline": None if (world > 1 or args.no_cpu_baseline) else cpu_baseline_training(workload, config),
    }


def bench_dataset(args, world, rank, device, distributed):
    """Dataset-creation hot loop (BASELINE.json configs[2], per GPU): seeded CPU latents -> Generator.forward with
    activations and fresh device noise -> nearest k-means centre maps for the reference config's four activation
    keys (8, 9, 12, 13) -> uint8 images, all on the device; PNG encoding / file IO is not part of the step."""
    import torch.distributed as dist
    import sis_hip
    from segmentation.gan_local_edit.factor_catalog import FactorCatalog
    from utils.dataset_creation import label_and_encode, seeded_latents
    batch = args.batch or BATCH
    g = build_generator(device)
    rng = np.random.RandomState(7)
    chans = {8: 512, 9: 512, 12: 128, 13: 128}
    catalogs = {k: FactorCatalog(cluster_centers=rng.randn(24, c).astype(np.float32)) for k, c in chans.items()}
    torch.random.manual_seed(1 + rank)

    def one_batch():
        with torch.no_grad():
            z = seeded_latents(batch, g.style_dim, device).to(device, non_blocking=True)   # pinned: the host runs a batch ahead
            image, acts = g([z], noise=g.make_noise(), return_intermediate_activations=True)
            return label_and_encode(image, acts, catalogs)  # side stream: overlaps the next batch's first layers

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_batch()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_batch()
    fence()
    elapsed = time.perf_counter() - t0
    assert out[0].dtype == torch.uint8
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank != 0:
        return None
    images = batch * args.steps * world
    return {"metric": METRIC, "value": round(images / elapsed, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "create_dataset_for_segmentation hot loop: Generator(256) forward + 4 k-means label "
                                   "maps (24 centres, layers 8/9/12/13) + uint8 images, no file IO "
                                   "(BASELINE.json configs[2], per-GPU shard)", "batch_per_gpu": batch,
                       "image_size": SIZE, "parallelism": f"image-id shards x{world}, no collective"},
            "roofline": {"kernel": "whole loop body", "bound": "mfma", "achieved": round(
                90.24e9 * images / elapsed / 1e12 / world, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(90.24e9 * images / elapsed / 1e12 / world / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None}}


def measured_traffic(kernel, workload=None):
    """HBM bytes per launch of ``kernel`` from the committed PMC passes (profiles/*traffic*.json, written by
    tools/pmc_traffic.sh on the GPU box: counters cannot be collected from inside the timed process)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*traffic*.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        data = json.load(f)
    table = data.get("kernels", {}) if workload is None else data.get("training", {}).get(workload, {})
    hits = [(v.get("launches", 0), v["traffic_bytes"]) for name, v in table.items()
            if name == kernel or name.startswith(kernel.rstrip(">") + ",") or name.startswith(kernel + "<")]
    if not hits:
        return None, None
    return max(hits)[1], os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__)))  # template instances: the most launched


def bench_gan(args, world, rank, device, distributed):
    """StyleGAN2 training images/s (reference train_stylegan_2.py:57-124 +
    configs/stylegan/stylegan_256px.yaml: 256^2, batch 24, lr 1e-3, lazy regularisers every 4 / 16 iterations): one step
    = one ``Stylegan2Updater.update_core`` (D step, [R1], G step, [path length], g_ema average) on real images resident
    in HBM.  Run with --steps a multiple of 16 so that the timed region holds whole regulariser cycles."""
    import torch.distributed as dist
    from networks.stylegan2.model import Discriminator, Generator
    from updater.stylegan_2_updater import Stylegan2Updater
    batch = args.batch or 24
    if args.miopen_search:
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(rank)
    g, g_ema = Generator(256, 512, 8, channel_multiplier=2).to(device), Generator(256, 512, 8, channel_multiplier=2).to(device)
    g_ema.eval()
    d = Discriminator(256, channel_multiplier=2).to(device)
    g_ratio, d_ratio = 4 / 5, 16 / 17
    opts = {"generator": torch.optim.Adam(g.parameters(), lr=1e-3 * g_ratio, betas=(0.0, 0.99 ** g_ratio)),
            "discriminator": torch.optim.Adam(d.parameters(), lr=1e-3 * d_ratio, betas=(0.0, 0.99 ** d_ratio))}
    if distributed:
        from torch.nn.parallel import DistributedDataParallel as DDP
        g = DDP(g, device_ids=[device.index], broadcast_buffers=False, output_device=device.index)
        d = DDP(d, device_ids=[device.index], broadcast_buffers=False, output_device=device.index)
    images = torch.rand(batch, 3, 256, 256, device=device) * 2 - 1

    def batches():
        while True:
            yield {"image": images.clone()}

    updater = Stylegan2Updater(iterators={"images": batches()}, networks={"generator": g, "discriminator": d}, optimizers=opts,
                               device=device, g_ema=g_ema, latent_size=512, style_mixing_prob=0.9,
                               regularization_options={"g_interval": 4, "d_interval": 16, "r1_weight": 10, "path_reg_weight": 2},
                               freeze_stochastic_noise_layers=[0, 1, 2, 3, 4, 5])
    updater.accumulate(g, 0)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        updater.update()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        updater.update()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank != 0:
        return None
    n_images = batch * args.steps * world
    return {
        "metric": METRIC, "value": round(n_images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"StyleGAN2 GAN training iteration (D, lazy R1 /16, G, lazy path length /4, g_ema), 256x256, "
                               f"batch {batch} per GPU, Adam (SURVEY.md 8(f) row 4; not a BASELINE.json config)",
                   "batch_per_gpu": batch, "image_size": 256, "parallelism": f"dp{world}",
                   "modconv": "grouped" if os.environ.get("SIS_MODCONV_GROUPED", "0") == "1" else "shared-weight",
                   "winograd": os.environ.get("SIS_GAN_WINOGRAD", "1") != "0", "miopen_search": bool(args.miopen_search)},
        "roofline": None,
    }


def bench_synthesis(args, world, rank, device, distributed):
    """BASELINE.json configs[1]: the headline ``value``."""
    import torch.distributed as dist
    batch = args.batch or BATCH
    g = build_generator(device)
    z, noise = synth_inputs(g, batch, device, seed=1 + rank)

    out = None
    for _ in range(args.warmup):
        out = step(g, z, noise)  # same hold-previous-outputs pattern as the timed loop: the caching allocator
        # reaches its high-water mark here, not in the first timed steps (a hipMalloc of several GB costs ~10 ms)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(g, z, noise)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(out[0]).all()
    # SURVEY.md §8(d) config 2 asks for both modes: (ii) with the 14 activations returned is `value` above (what
    # dataset creation calls); (i) image only is timed here the same way (same kernels, nothing retained).
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        with torch.no_grad():
            img_only = g([z], noise=noise)
    fence()
    elapsed_image_only = time.perf_counter() - t1
    del img_only
    if distributed:
        t = torch.tensor([elapsed_image_only], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_image_only = t.item()
    if rank != 0:
        return None

    n_gpus = world
    total_images = batch * args.steps * n_gpus
    prof_steps = min(args.steps, 5)
    agg = kernel_profile(g, z, noise, prof_steps)
    dom_name = max(agg, key=lambda k: agg[k]["ms"])
    dom = agg[dom_name]
    per_launch_ms = dom["ms"] / dom["launches"]
    algorithmic = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    # The Winograd F(2x2,3x3) kernel executes 16 of the 36 multiplies of the direct form per 2x2 output tile.  SURVEY.md
    # §8(d) defines the unit work as direct-convolution FLOPs ("algorithmic_*" below, which can exceed the peak); the
    # roofline fraction proper is what the MFMA pipe EXECUTES: achieved = algorithmic * 16/36, frac = achieved / peak <= 1.
    dom_ratio = exec_ratio(dom_name)
    executed = algorithmic * dom_ratio
    traffic, traffic_src = measured_traffic(dom_name)
    roofline = {"kernel": dom_name, "bound": "mfma", "achieved": round(executed, 2), "peak": PEAK_MFMA_F32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(executed / PEAK_MFMA_F32_TFLOPS, 4),
                "note": "achieved / frac = EXECUTED fp32 MFMA rate (Winograd: 16/36 of the direct-form FLOPs); "
                        "algorithmic_* = direct-form 2*MAC FLOPs of SURVEY.md 8(d) per launch / launch time",
                "algorithmic_tflops": round(algorithmic, 2),
                "algorithmic_frac": round(algorithmic / PEAK_MFMA_F32_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "bytes per launch (HBM-side, PMC)", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                "launches_per_step": dom["launches"] // prof_steps,
                "avg_launch_ms": round(per_launch_ms, 4),
                "flops_per_launch": dom["flops"] / dom["launches"],
                "executed_flops_per_launch": dom["flops"] / dom["launches"] * dom_ratio,
                "kernels": {k: {"ms_per_step": round(v["ms"] / prof_steps, 4),
                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                "executed_tflops": round(v["flops"] * exec_ratio(k) / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}}
    result = {
        "metric": METRIC, "value": round(total_images / elapsed, 2), "unit": "images/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"StyleGAN2 Generator(256,512,8,cm=2).forward, batch {batch} per GPU, explicit noise, "
                               "return_intermediate_activations=True (BASELINE.json configs[1])",
                   "batch_per_gpu": batch, "image_size": SIZE, "parallelism": f"replicated x{n_gpus}, "
                   "images sharded, no collective",
                   "images_per_s_image_only": round(total_images / elapsed_image_only, 2)},
        "roofline": roofline,
    }
    if n_gpus == 1 and not args.no_cpu_baseline:
        cb, img_cpu, (z4, noise4) = cpu_baseline(g, seed=1234)
        with torch.no_grad():
            img_gpu, _ = g([z4.to(device)], noise=[n.to(device) for n in noise4])
        cb["gpu_vs_cpu_image_max_rel_err"] = float(
            (img_gpu.cpu() - img_cpu).abs().max() / img_cpu.abs().max())
        result["cpu_baseline"] = cb
    return result


DETAIL_KEYS = ("kernels", "own_kernels_eager_iteration")   # per-kernel tables: detail file, never the JSON line


def seg_summary(sub):
    """The compact record of one segmentation-training leg: what the driver's parsed record has room for."""
    roof, dp, lib = sub["roofline"], sub.get("data_parallel_rehearsal") or {}, sub.get("library_ms_per_step") or {}
    cpu = sub.get("cpu_baseline") or {}
    return {"images_per_s": sub["value"], "ms_per_step": sub["ms_per_step"], "dtype": sub["dtype"], "steps": sub["steps"],
            "batch_per_gpu": sub["config"]["batch_per_gpu"], "image_size": sub["config"]["image_size"],
            "hip_graph": sub["config"]["hip_graph"],
            "roofline": {"bound": roof["bound"], "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"],
                         "frac": roof["frac"], "algorithmic_frac": roof["algorithmic_frac"],
                         "dominant_own_kernel": roof["dominant_own_kernel"], "traffic": roof["traffic"]},
            "cpu_baseline": {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")} if cpu else None,
            "cpu_baseline_batch": cpu.get("batch"),
            "library_ms_per_step": lib.get("library_ms"), "own_ms_per_step": lib.get("own_ms"),
            "library_fallback_ops": sum((sub.get("library_calls_per_step") or {}).get("fallback", {}).values()),
            "dp_rehearsal_graph_ms_per_step": dp.get("graph_ms_per_step"), "dp_rehearsal_eager_ms_per_step": dp.get("eager_ms_per_step"),
            "dp_direct_rccl": dp.get("direct_rccl"), "dp_gradients_written_in_place": dp.get("gradients_written_in_place")}


def attach_seg_train(result, seg):
    """``--workload all``: the two (three) training legs ride on the synthesis line three times over, because the driver's
    parsed record keeps `config`, `roofline` and `cpu_baseline` as FLAT objects (nested tables are dropped, strings cut at
    128 characters) plus the END of stdout: (1) flat ``seg_<leg>_*`` keys inside those three objects, (2) the compact nested
    ``seg_train`` object as the LAST key of the line, (3) the full per-kernel tables in the detail file (``emit``)."""
    result["_detail"] = {"seg_train": seg}
    summary = {k: seg_summary(v) for k, v in seg.items()}
    for leg, s in summary.items():
        c, r = result["config"], result["roofline"]
        c[f"seg_{leg}_images_per_s"], c[f"seg_{leg}_ms_per_step"] = s["images_per_s"], s["ms_per_step"]
        c[f"seg_{leg}_dtype"], c[f"seg_{leg}_batch_per_gpu"], c[f"seg_{leg}_hip_graph"] = s["dtype"], s["batch_per_gpu"], s["hip_graph"]
        c[f"seg_{leg}_library_ms_per_step"] = s["library_ms_per_step"]
        c[f"seg_{leg}_dp_rehearsal_graph_ms_per_step"] = s["dp_rehearsal_graph_ms_per_step"]
        for k in ("achieved", "peak", "frac", "algorithmic_frac", "dominant_own_kernel", "traffic"):
            r[f"seg_{leg}_{k}"] = s["roofline"][k]
        if s["cpu_baseline"] and result.get("cpu_baseline"):
            result["cpu_baseline"][f"seg_{leg}_value"] = s["cpu_baseline"]["value"]
            result["cpu_baseline"][f"seg_{leg}_sample_batch"] = s["cpu_baseline_batch"]
    result["seg_train"] = summary


def emit(result, args):
    """ONE compact JSON line on stdout (< 6 KB: tests/test_bench_launch_gpu.py); every per-kernel table goes to
    ``gpurun_out/bench_detail_<workload>.json`` (scratch; the judged copies are committed under profiles/)."""
    detail = result.pop("_detail", {})
    for obj in (result.get("roofline"),):
        if isinstance(obj, dict):
            for k in DETAIL_KEYS:
                if k in obj:
                    detail.setdefault("roofline_tables", {})[k] = obj.pop(k)
    if isinstance(result.get("library_ms_per_step"), dict):
        detail["top_library_kernels_ms"] = result["library_ms_per_step"].pop("top_library_kernels_ms", None)
    if detail:
        path = os.path.join(ROOT, "gpurun_out", f"bench_detail_{args.workload}.json")
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(detail, f, indent=1)
            print(f"bench.py: per-kernel tables -> {os.path.relpath(path, ROOT)}", file=sys.stderr)
        except OSError as err:
            print(f"bench.py: could not write the detail file ({err!r})", file=sys.stderr)
    if "seg_train" in result:          # last key of the line: what the tail of stdout shows
        result["seg_train"] = result.pop("seg_train")
    print(json.dumps(result))



def launch_ranks(n):
    """One child process group of ``n`` ranks through ``python -m torch.distributed.run`` (one rank per GPU, rendezvous on
    127.0.0.1, port chosen by the launcher), this script and its arguments unchanged.  The parent never initialises the GPU; the children's
    stdout / stderr pass straight through (rank 0 prints the one JSON line).  Returns the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    # --standalone: the launcher binds its own rendezvous port (no pick-then-release race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=["all", "synthesis", "emanet", "transunet", "dataset", "gan"],
                    help="all (default, what the driver runs): the synthesis headline + both segmentation training steps on "
                         "one JSON line")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"], help="training workloads: override the config's amp")
    ap.add_argument("--no-dp-rehearsal", dest="dp_rehearsal", action="store_false",
                    help="training workloads at N = 1: skip the data-parallel rehearsal over a world-size-1 RCCL communicator")
    ap.add_argument("--miopen-search", action="store_true",
                    help="training workloads: MIOpen solver search for the library convolutions (minutes at start-up)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: start the N ranks ourselves (the reference spawns its own workers too,
        # train.py:185-187), BEFORE anything in this process touches the GPU, relay rank 0's JSON line and exit with the
        # launcher's code.  Under torch.distributed.run (WORLD_SIZE set) this branch is skipped.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (the product path has no CPU fallback)")
    # Rehearsal switches (one-GPU boxes): SIS_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and SIS_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); the driver's multi-GPU runs use neither.
    if os.environ.get("SIS_BENCH_SHARE_GPU", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SIS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    # (N = 1: the one-rank RCCL communicator of the data-parallel rehearsal is created right before each rehearsal and destroyed
    # right after it -- single_rank_communicator below.  With it alive from the start, every bare measurement of the process ran
    # 0.5-1.3 % slower on the same box: synthesis 14.42 -> 14.61 ms per step, tools/ab_bench_modes.sh.)

    if args.workload == "dataset":
        result = bench_dataset(args, world, rank, device, distributed)
    elif args.workload == "gan":
        result = bench_gan(args, world, rank, device, distributed)
    elif args.workload in ("emanet", "transunet"):
        result = bench_training(args, args.workload, world, rank, device, distributed)
    else:
        result = bench_synthesis(args, world, rank, device, distributed)
        if args.workload == "all":
            # The metric names two numbers: `value` stays the configs[1] synthesis rate; the segmentation training steps of
            # configs[3] (EMANet-50, 256^2, B=16, fp32) and configs[4] (TransUNet R50-ViT-B/16, 512^2, B=8, bf16) are timed
            # next in the same process with the same K, each with its own roofline and CPU baseline.
            import copy
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            seg = {}
            # (transunet_f32: SURVEY.md 8(d) config 5 asks for "bf16 and an fp32 parity run" -- the reference's own precision,
            # fewer timed steps, no CPU baseline of its own: the oracle step is the bf16 entry's baseline too)
            for key, workload, dtype in (("emanet", "emanet", "f32"), ("transunet_bf16", "transunet", "bf16"),
                                         ("transunet_f32", "transunet", "f32")):
                sub_args = copy.copy(args)
                sub_args.batch, sub_args.dtype, sub_args.miopen_search = 0, dtype, False
                if key == "transunet_f32":
                    sub_args.steps, sub_args.no_cpu_baseline, sub_args.dp_rehearsal = max(4, args.steps // 2), True, False
                sub = bench_training(sub_args, workload, world, rank, device, distributed)
                gc.collect()
                torch.cuda.empty_cache()
                if sub is not None:
                    seg[key] = sub
            if result is not None:
                attach_seg_train(result, seg)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    else:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    if result is not None:
        emit(result, args)


if __name__ == "__main__":
    main()
