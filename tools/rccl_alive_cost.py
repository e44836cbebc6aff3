"""What does a live world-size-1 RCCL communicator cost the synthesis step?  One process: rate before the communicator exists,
with it alive, after it is destroyed; per phase the step time and the HIP-event time of the dominant kernel (bench.kernel_profile).
usage: python tools/rccl_alive_cost.py   (GPU box)"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
COMM_FIRST = len(sys.argv) > 1 and sys.argv[1] == "first"   # the communicator before the generator and its buffers exist (bench.py's old order)


def make_comm():
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)


if COMM_FIRST:
    make_comm()
g = bench.build_generator(dev)
z, noise = bench.synth_inputs(g, bench.BATCH, dev, seed=1)


def phase(tag):
    out = None
    for _ in range(5):
        out = bench.step(g, z, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = bench.step(g, z, noise)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    agg = bench.kernel_profile(g, z, noise, 5)
    dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
    own = sum(a["ms"] for a in agg.values()) / 5
    print(f"{tag:28s} {ms:7.3f} ms/step   {dom[0]} {dom[1]['ms'] / dom[1]['launches']:.4f} ms/launch   sum of kernels {own:.3f} ms/step", flush=True)
    del out


if COMM_FIRST:
    phase("communicator made first")
    phase("communicator made first (2)")
    import torch.distributed as dist  # noqa: E402
    dist.destroy_process_group()
    phase("destroyed")
else:
    phase("no communicator")
    phase("no communicator (2)")
    make_comm()
    import torch.distributed as dist  # noqa: E402
    phase("group initialised")
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    phase("after one all_reduce")
    dist.destroy_process_group()
    phase("communicator destroyed")
