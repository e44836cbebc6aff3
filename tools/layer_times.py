"""Per-LAUNCH listing of the own kernels of one eager training iteration (sis_hip.set_profiler: HIP events around every launch):
kernel, algorithmic bytes, FLOPs, microseconds, GB/s, TFLOP/s -- for finding which SHAPES of a kernel are far from its roofline.
usage: layer_times.py emanet|transunet [substring filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets sys.path for the package)
import torch  # noqa: E402
import yaml  # noqa: E402

workload = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
from training_builder.train_builder_selection import get_train_builder_class  # noqa: E402
from utils.synthetic_data import SyntheticSegmentationLoader  # noqa: E402
import sis_hip  # noqa: E402
device = torch.device("cuda:0")
torch.cuda.set_device(device)
config = yaml.safe_load(open(os.path.join(ROOT, "synthesis-in-style_amd", bench.SEG_CONFIG[workload])))
config["fine_tune"] = None
config["hip_graph"] = False
loader = SyntheticSegmentationLoader(config["batch_size"], config["image_size"], config["num_classes"], seed=1234, device=device)
torch.manual_seed(0)
updater = get_train_builder_class(config)(config, loader, None, rank=0, world_size=1).get_updater()
if getattr(updater, "_step_graph", None) is not None:
    updater._step_graph.requested = False
for _ in range(2):
    updater.update()
records = []
sis_hip.set_profiler(records)
updater.update()
torch.cuda.synchronize()
sis_hip.set_profiler(None)
agg = {}
for name, flops, nbytes, e0, e1 in records:
    if flt not in name:
        continue
    us = e0.elapsed_time(e1) * 1e3
    key = (name, int(nbytes), int(flops))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += us
print(f"{'kernel':58s} {'MB':>8s} {'GFLOP':>8s} {'n':>3s} {'us/launch':>9s} {'GB/s':>7s} {'TF':>6s} {'total us':>9s}")
for (name, nbytes, flops), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    per = us / n
    print(f"{name[:58]:58s} {nbytes / 1e6:8.1f} {flops / 1e9:8.1f} {n:3d} {per:9.1f} {nbytes / per / 1e3:7.0f} {flops / per / 1e6:6.1f} {us:9.0f}")
