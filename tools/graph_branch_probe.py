"""Do the parallel branches of a captured hipGraph run concurrently at replay?  Two spin kernels, one on the capture stream and one
on a forked side stream, joined; replayed and timed.  usage: python tools/graph_branch_probe.py [n_branches=2]"""
import sys

import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
CYC = 20_000_000
sides = [torch.cuda.Stream(device=dev) for _ in range(N - 1)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


alone = min(timed(lambda: torch.cuda._sleep(CYC)) for _ in range(3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    for s in sides:
        s.wait_stream(cur)
    torch.cuda._sleep(CYC)
    for s in sides:
        with torch.cuda.stream(s):
            torch.cuda._sleep(CYC)
    for s in sides:
        cur.wait_stream(s)
g.replay()
both = min(timed(g.replay) for _ in range(3))
print(f"one spin kernel: {alone:.3f} ms; graph of {N} parallel spin kernels: {both:.3f} ms = {both / alone:.2f} x")
