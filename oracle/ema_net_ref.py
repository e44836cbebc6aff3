"""ORACLE (test infrastructure, not product): functional CPU restatement of the reference EMANet
training step, over a plain state_dict.

Follows /root/reference/stylegan_code_finder:
  networks/ema_net/network.py   Bottleneck :18-56, ResNet (deep stem, output stride 8, dilation grids) :59-148,
                                ConvBNReLU :169-184, EMAU :187-264, EMANet.forward :296-311,
                                CrossEntropyLoss2d :319-327, norm layer = BatchNorm(momentum 3e-4) :14-15
  networks/ema_net/bn_lib/nn/modules/batchnorm.py:51-56   (under DDP the "synchronized" BN is plain F.batch_norm)
  networks/ema_net/utils.py:7-21                          SGD parameter groups 1x / 1y / 2x
  updater/segmentation_updater.py:47-73                   step order: fwd -> mu EMA -> mean -> zero_grad/backward/step
  training_builder/ema_net_train_builder.py:27-48         SGD(momentum) with (lr, wd) / (lr, 0) / (2 lr, 0)

Written as functions over ``{name: tensor}`` (the product is nn.Module based), so agreement is evidence.
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

BN_MOM = 3e-4
BN_EPS = 1e-5
LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def _bn_keys(prefix, c):
    return [(f"{prefix}.weight", (c,)), (f"{prefix}.bias", (c,)), (f"{prefix}.running_mean", (c,)),
            (f"{prefix}.running_var", (c,)), (f"{prefix}.num_batches_tracked", ())]


def _stage_plan(n_layers):
    """(prefix, inplanes, planes, stride, dilation of conv2, has_downsample) for every bottleneck, output
    stride 8 (network.py:82-86,101-131): layer3 dilation 2, layer4 dilation 4 with grids [1,2,4]."""
    blocks = LAYERS[n_layers]
    plan = []
    inplanes = 128
    spec = [(64, blocks[0], 1, 1, None), (128, blocks[1], 2, 1, None), (256, blocks[2], 1, 2, None),
            (512, blocks[3], 1, 4, [1, 2, 4])]
    for li, (planes, n, stride, dilation, grids) in enumerate(spec):
        grids = grids or [1] * n
        first_dil = 1 if dilation in (1, 2) else 2
        down = stride != 1 or inplanes != planes * 4
        plan.append((f"extractor.{4 + li}.0", inplanes, planes, stride, first_dil, down))
        inplanes = planes * 4
        for i in range(1, n):
            plan.append((f"extractor.{4 + li}.{i}", inplanes, planes, 1, dilation * grids[i], False))
    return plan


def state_dict_schema(n_layers=50, num_classes=3):
    """Ordered (name, shape) list of EMANet(num_classes, n_layers).state_dict() (353 entries for 50 layers)."""
    out = [("extractor.0.0.weight", (64, 3, 3, 3))] + _bn_keys("extractor.0.1", 64)
    out += [("extractor.0.3.weight", (64, 64, 3, 3))] + _bn_keys("extractor.0.4", 64)
    out += [("extractor.0.6.weight", (128, 64, 3, 3))] + _bn_keys("extractor.1", 128)
    for prefix, cin, planes, stride, dil, down in _stage_plan(n_layers):
        out += [(f"{prefix}.conv1.weight", (planes, cin, 1, 1))] + _bn_keys(f"{prefix}.bn1", planes)
        out += [(f"{prefix}.conv2.weight", (planes, planes, 3, 3))] + _bn_keys(f"{prefix}.bn2", planes)
        out += [(f"{prefix}.conv3.weight", (planes * 4, planes, 1, 1))] + _bn_keys(f"{prefix}.bn3", planes * 4)
        if down:
            out += [(f"{prefix}.downsample.0.weight", (planes * 4, cin, 1, 1))] + _bn_keys(f"{prefix}.downsample.1",
                                                                                          planes * 4)
    out += [("fc0.conv.weight", (512, 2048, 3, 3))] + _bn_keys("fc0.bn", 512)
    out += [("emau.mu", (1, 512, 64)), ("emau.conv1.weight", (512, 512, 1, 1)), ("emau.conv1.bias", (512,)),
            ("emau.conv2.0.weight", (512, 512, 1, 1))] + _bn_keys("emau.conv2.1", 512)
    out += [("fc1.0.conv.weight", (256, 512, 3, 3))] + _bn_keys("fc1.0.bn", 256)
    out += [("fc2.weight", (num_classes, 256, 1, 1)), ("fc2.bias", (num_classes,))]
    return out


def seeded_state_dict(n_layers=50, num_classes=3, seed=0, residual_scale=1.0):
    """Platform-independent synthetic checkpoint (numpy RandomState, schema order) with the reference's init
    scales: conv ~ N(0, sqrt(2/(k*k*Cout))) (network.py:91-93), BN weight 1 / bias 0 perturbed by 0.1 so the
    affine path is exercised, running stats 0 / 1, mu ~ N(0, sqrt(2/k)) l2-normalised over C (:199-202).
    ``residual_scale`` multiplies the last batch norm (bn3) of every bottleneck: 0.1 is the usual small / zero
    initialisation of residual branches.  With 1.0 the 16-block random network is chaotic (a 1e-5 input perturbation
    moves every backbone gradient by 15 % in L2, measured on this oracle); with 0.1 by 1-2 %, the floor a ReLU network
    allows (each unit whose pre-activation crosses zero flips a whole gradient element)."""
    rng = np.random.RandomState(seed)
    sd = {}
    for name, shape in state_dict_schema(n_layers, num_classes):
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        elif name == "emau.mu":
            mu = torch.from_numpy(rng.standard_normal(shape)).float() * math.sqrt(2.0 / shape[2])
            sd[name] = mu / (1e-6 + mu.norm(dim=1, keepdim=True))
        elif len(shape) == 4:
            std = math.sqrt(2.0 / (shape[2] * shape[3] * shape[0]))
            sd[name] = (torch.from_numpy(rng.standard_normal(shape)) * std).float()
        elif name.endswith(".weight"):  # BN scale
            sd[name] = (1 + 0.1 * torch.from_numpy(rng.standard_normal(shape))).float()
        else:  # BN shift / conv bias
            sd[name] = (0.1 * torch.from_numpy(rng.standard_normal(shape))).float()
        if residual_scale != 1.0 and (name.endswith("bn3.weight") or name.endswith("bn3.bias")):
            sd[name] = sd[name] * residual_scale
    return sd


def seeded_batch(batch, size, num_classes, seed):
    """Batch contract of the reference loaders (data/segmentation_dataset.py:60-63): images f32 [B,3,S,S] in
    [-1,1], labels int64 [B,1,S,S] in [0, C)."""
    rng = np.random.RandomState(seed)
    images = torch.from_numpy(rng.uniform(-1, 1, (batch, 3, size, size))).float()
    labels = torch.from_numpy(rng.randint(0, num_classes, (batch, 1, size, size))).long()
    return {"images": images, "segmented": labels}


class _Ctx:
    def __init__(self, sd, training):
        self.sd, self.training, self.new_stats = sd, training, {}

    def bn(self, x, prefix):
        sd = self.sd
        rm, rv = sd[f"{prefix}.running_mean"].clone(), sd[f"{prefix}.running_var"].clone()
        y = F.batch_norm(x, rm, rv, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], self.training, BN_MOM, BN_EPS)
        self.new_stats[f"{prefix}.running_mean"], self.new_stats[f"{prefix}.running_var"] = rm, rv
        return y


def _l2norm(t, dim):
    return t / (1e-6 + t.norm(dim=dim, keepdim=True))


def forward(sd, img, lbl=None, n_layers=50, training=True, stage_num=3, ignore_label=255, size=None):
    """Returns (loss[B], mu[B,512,64], new_running_stats) when training with labels, else (pred, new_stats)."""
    c = _Ctx(sd, training)
    x = F.conv2d(img, sd["extractor.0.0.weight"], stride=2, padding=1)
    x = F.relu(c.bn(x, "extractor.0.1"))
    x = F.relu(c.bn(F.conv2d(x, sd["extractor.0.3.weight"], padding=1), "extractor.0.4"))
    x = F.relu(c.bn(F.conv2d(x, sd["extractor.0.6.weight"], padding=1), "extractor.1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for prefix, cin, planes, stride, dil, down in _stage_plan(n_layers):
        res = x
        y = F.relu(c.bn(F.conv2d(x, sd[f"{prefix}.conv1.weight"]), f"{prefix}.bn1"))
        y = F.conv2d(y, sd[f"{prefix}.conv2.weight"], stride=stride, padding=dil, dilation=dil)
        y = F.relu(c.bn(y, f"{prefix}.bn2"))
        y = c.bn(F.conv2d(y, sd[f"{prefix}.conv3.weight"]), f"{prefix}.bn3")
        if down:
            res = c.bn(F.conv2d(x, sd[f"{prefix}.downsample.0.weight"], stride=stride), f"{prefix}.downsample.1")
        x = F.relu(y + res)
    x = F.relu(c.bn(F.conv2d(x, sd["fc0.conv.weight"], padding=1), "fc0.bn"))
    # EMAU (network.py:219-249)
    idn = x
    x = F.conv2d(x, sd["emau.conv1.weight"], sd["emau.conv1.bias"])
    b, ch, h, w = x.shape
    xf = x.view(b, ch, h * w)
    mu = sd["emau.mu"].repeat(b, 1, 1)
    with torch.no_grad():
        for _ in range(stage_num):
            z = F.softmax(torch.bmm(xf.permute(0, 2, 1), mu), dim=2)
            z_ = z / (1e-6 + z.sum(dim=1, keepdim=True))
            mu = _l2norm(torch.bmm(xf, z_), dim=1)
    x = F.relu(mu.matmul(z.permute(0, 2, 1)).view(b, ch, h, w))
    x = c.bn(F.conv2d(x, sd["emau.conv2.0.weight"]), "emau.conv2.1")
    x = F.relu(x + idn)
    x = F.relu(c.bn(F.conv2d(x, sd["fc1.0.conv.weight"], padding=1), "fc1.0.bn"))
    # Dropout2d(p=0.1) of fc1 is stochastic (device RNG): parity runs use p = 0
    x = F.conv2d(x, sd["fc2.weight"], sd["fc2.bias"])
    pred = F.interpolate(x, size=size or img.shape[-2:], mode="bilinear", align_corners=True)
    if training and lbl is not None:
        nll = F.nll_loss(F.log_softmax(pred, dim=1), lbl, ignore_index=ignore_label, reduction="none")
        return nll.mean(dim=2).mean(dim=1), mu, c.new_stats
    return pred, c.new_stats


def param_groups(sd):
    """networks/ema_net/utils.py:7-21 over a state_dict: conv weights / BN weights / all biases."""
    g1x = [k for k, v in sd.items() if v.dim() == 4]
    g1y = [k for k in sd if k.endswith(".weight") and sd[k].dim() == 1]
    g2x = [k for k in sd if k.endswith(".bias")]
    return g1x, g1y, g2x


def train_step(sd, momentum_buffers, batch, lr=0.009, lr_mom=0.9, weight_decay=1e-4, em_mom=0.9, n_layers=50):
    """One EMANetUpdater.update_core (segmentation_updater.py:47-73) with torch.optim.SGD semantics
    (first step: buf = grad + wd * p; p -= lr * buf).  Updates ``sd`` / ``momentum_buffers`` in place and
    returns (loss_mean, per-sample loss, mu, grads)."""
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k and k != "emau.mu"]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    loss, mu, new_stats = forward(work, batch["images"], batch["segmented"].squeeze(1), n_layers=n_layers)
    with torch.no_grad():
        sd["emau.mu"] = sd["emau.mu"] * em_mom + mu.mean(dim=0, keepdim=True) * (1 - em_mom)
    total = loss.mean()
    # emau.conv1 only feeds the no_grad EM iterations (network.py:229-240): it gets no gradient, and SGD skips it
    grads = dict(zip(names, torch.autograd.grad(total, [leaves[k] for k in names], allow_unused=True)))
    g1x, g1y, g2x = param_groups(sd)
    with torch.no_grad():
        for group, g_lr, wd in ((g1x, lr, weight_decay), (g1y, lr, 0.0), (g2x, 2 * lr, 0.0)):
            for k in group:
                if grads[k] is None:
                    continue
                d = grads[k] + wd * sd[k] if wd else grads[k].clone()
                buf = momentum_buffers.get(k)
                buf = d if buf is None else buf * lr_mom + d
                momentum_buffers[k] = buf
                sd[k] = sd[k] - g_lr * buf
        for k, v in new_stats.items():
            sd[k] = v.detach()
    return total.detach(), loss.detach(), mu.detach(), grads
