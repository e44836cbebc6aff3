"""Which tensor copies does one eager training iteration issue?  torch.profiler with input shapes: aten::copy_ / contiguous / cat /
clone rows sorted by device time.  usage: find_copies.py emanet|transunet"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

workload = sys.argv[1]
from training_builder.train_builder_selection import get_train_builder_class  # noqa: E402
from utils.synthetic_data import SyntheticSegmentationLoader  # noqa: E402
device = torch.device("cuda:0")
torch.cuda.set_device(device)
config = yaml.safe_load(open(os.path.join(ROOT, "synthesis-in-style_amd", bench.SEG_CONFIG[workload])))
config["fine_tune"] = None
config["hip_graph"] = False
loader = SyntheticSegmentationLoader(config["batch_size"], config["image_size"], config["num_classes"], seed=1234, device=device)
torch.manual_seed(0)
updater = get_train_builder_class(config)(config, loader, None, rank=0, world_size=1).get_updater()
for _ in range(3):
    updater.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    updater.update()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if ev.key in ("aten::copy_", "aten::cat", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul") \
            and getattr(ev, "device_time_total", 0) > 0:
        rows.append((ev.device_time_total, ev.count, ev.key, str(ev.input_shapes)[:90], [s for s in ev.stack if "site-packages" not in s and "dist-packages" not in s][:3]))
rows.sort(reverse=True)
for t, n, key, shapes, stack in rows[:40]:
    print(f"{t:9.1f} us {n:3d}x {key:16s} {shapes}")
    for s in stack:
        print("            ", s[-110:])
