"""How long does the host take to ISSUE one generator step (no device sync inside)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch
import bench
dev = torch.device("cuda:0")
g = bench.build_generator(dev)
z, noise = bench.synth_inputs(g, 32, dev, 1)
for _ in range(3):
    bench.step(g, z, noise)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    outs = []
    for _ in range(5):
        outs.append(bench.step(g, z, noise)[0])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e3*(t1-t0)/5:.2f} ms/step, total {1e3*(t2-t0)/5:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    bench.step(g, z, noise)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
