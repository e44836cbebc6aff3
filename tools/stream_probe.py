"""Which of torch's pool streams run CONCURRENTLY with the current stream?  (HIP maps streams onto a few hardware queues; two
streams on one queue run their kernels one after the other.)  usage: python tools/stream_probe.py [first]   (first: after a
world-size-1 RCCL communicator has been created, bench.py's old order)"""
import os
import socket
import sys

import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if len(sys.argv) > 1 and sys.argv[1] == "first":
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
main = torch.cuda.current_stream(dev)
CYC = 20_000_000


def timed(cand):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(main)
    if cand is not None:
        cand.wait_event(e0)
    torch.cuda._sleep(CYC)
    if cand is not None:
        with torch.cuda.stream(cand):
            torch.cuda._sleep(CYC)
        main.wait_stream(cand)
    e1.record(main)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1)


timed(None)
alone = min(timed(None) for _ in range(3))
print(f"one spin kernel alone: {alone:.3f} ms")
for i in range(10):
    s = torch.cuda.Stream(device=dev)
    both = min(timed(s) for _ in range(2))
    print(f"stream {i} (id {s.stream_id}, handle {s.cuda_stream:#x}): both {both:.3f} ms = {both / alone:.2f} x  ->  {'CONCURRENT' if both < 1.5 * alone else 'serial'}")
for i in range(6):   # torch.distributed's ProcessGroupNCCL takes its stream from the HIGH-priority pool
    s = torch.cuda.Stream(device=dev, priority=-1)
    both = min(timed(s) for _ in range(2))
    print(f"high-priority stream {i} (id {s.stream_id}): both {both:.3f} ms = {both / alone:.2f} x  ->  {'CONCURRENT' if both < 1.5 * alone else 'serial'}")
