"""Operator-level view of one eager training step (torch.profiler, shapes recorded): which ATen ops own the copy / cast /
reduce kernels the rocprofv3 breakdown shows.  usage: python tools/profile_ops.py transunet|emanet [filter]"""
import os, sys
os.environ["SIS_STEP_GRAPH"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd")); sys.path.insert(0, ROOT)
import torch, yaml
from torch.profiler import ProfilerActivity, profile
import bench
from training_builder.train_builder_selection import get_train_builder_class
from utils.synthetic_data import SyntheticSegmentationLoader

workload = sys.argv[1] if len(sys.argv) > 1 else "transunet"
config = yaml.safe_load(open(os.path.join(ROOT, "synthesis-in-style_amd", bench.SEG_CONFIG[workload])))
config["fine_tune"] = None
device = torch.device("cuda", 0)
loader = SyntheticSegmentationLoader(config["batch_size"], config["image_size"], config["num_classes"], seed=1234, device=device)
updater = get_train_builder_class(config)(config, loader, None, rank=0, world_size=1).get_updater()
for _ in range(3):
    updater.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    updater.update()
    torch.cuda.synchronize()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = [e for e in prof.key_averages(group_by_input_shape=True) if flt in e.key]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:int(os.environ.get("TOP", "60"))]:
    print(f"{e.self_device_time_total / 1e3:8.3f} ms {e.count:4d}  {e.key:40s} {str(e.input_shapes)[:150]}")
