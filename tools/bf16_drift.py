"""Where does the bf16 autocast path of TransUNet drift from fp32?  Relative L2 distance of every leaf module's output
between an fp32 run and a bf16-autocast run of the same product network, same weights, same batch (512^2, B = 2).
python tools/bf16_drift.py [train|eval]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "synthesis-in-style_amd")):
    sys.path.insert(0, p)
from oracle import ema_net_ref as E  # noqa: E402
from oracle import trans_u_net_ref as T  # noqa: E402


def main():
    from networks.trans_u_net.vit_seg_modeling import VIT_CONFIGS, VisionTransformer
    mode = sys.argv[1] if len(sys.argv) > 1 else "train"
    dev = torch.device("cuda:0")
    cfg = VIT_CONFIGS["R50-ViT-B_16"].copy()
    cfg.n_classes, cfg.n_skip = 3, 3
    cfg.patches.grid = (32, 32)
    cfg.transformer.dropout_rate = 0.0
    net = VisionTransformer(cfg, img_size=512, num_classes=3)
    sd = T.seeded_state_dict(512, 3, seed=3)
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0  # residual-branch scale (tests/test_trans_u_net_gpu.py::_vit_like)
    for k in sd:
        if ".gn3." in k:
            sd[k] = sd[k] * scale
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    net.train() if mode == "train" else net.eval()
    x = E.seeded_batch(2, 512, 3, seed=40)["images"].to(dev)
    store = {}

    def hook(name):
        def fn(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            if torch.is_tensor(o):
                store.setdefault(name, []).append(o.detach().float())
        return fn

    for name, mod in net.named_modules():
        if name and len(list(mod.children())) == 0 or name.endswith(("unit1", "unit2", "unit3", "unit4", "unit9")) or name in (
                "transformer.embeddings", "transformer.encoder", "decoder", "decoder.conv_more") or name.startswith("decoder.blocks.") and name.count(".") == 2:
            mod.register_forward_hook(hook(name))
    with torch.no_grad():
        net(x)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            net(x)
    print(f"{'module':60s} rel_l2   rms(ref)")
    for name, outs in store.items():
        if len(outs) != 2:
            continue
        a, b = outs
        if a.shape == b.shape:
            print(f"{name:60s} {((a - b).norm() / (a.norm() + 1e-20)).item():.4f}  {a.pow(2).mean().sqrt().item():.3g}")


if __name__ == "__main__":
    main()
