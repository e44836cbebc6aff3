"""ORACLE (test infrastructure, not product): functional CPU restatement of the reference TransUNet
(R50 + ViT-B/16 hybrid) training step over a plain state_dict.

Follows /root/reference/stylegan_code_finder:
  networks/trans_u_net/vit_seg_modeling_resnet_skip.py  StdConv2d :20-27, PreActBottleneck :40-75, ResNetV2 :114-162
  networks/trans_u_net/vit_seg_modeling.py              Attention :53-97, Mlp :100-122, Embeddings :125-168,
                                                        Block :171-190, Encoder :233-250, DecoderBlock :290-321,
                                                        DecoderCup :332-373, VisionTransformer.forward :393-399
  networks/trans_u_net/vit_seg_configs.py:6-62          ViT-B/16 + R50 (3,4,9), decoder (256,128,64,16), skips [512,256,64,0]
  networks/trans_u_net/utils.py:7-42                    DiceLoss
  updater/segmentation_updater.py:83-106                0.5 * CE + 0.5 * Dice(softmax), zero_grad / backward / step
  training_builder/trans_u_net_train_builder.py:39-40   SGD(lr, momentum, weight_decay) over all parameters
Dropout layers are stochastic (device RNG): parity runs use rate 0.
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

HIDDEN, MLP_DIM, HEADS, LAYERS = 768, 3072, 12, 12
RESNET_UNITS = (3, 4, 9)
DECODER = (256, 128, 64, 16)
SKIPS = (512, 256, 64, 0)  # n_skip = 3: the 4th skip is dropped (vit_seg_modeling.py:348-351)


def _units():
    """(prefix, cin, cout, cmid, stride) of every PreActBottleneck."""
    w = 64
    out = []
    spec = [(w, w * 4, w, 1), (w * 4, w * 8, w * 2, 2), (w * 8, w * 16, w * 4, 2)]
    for bi, ((cin, cout, cmid, stride), n) in enumerate(zip(spec, RESNET_UNITS)):
        for u in range(1, n + 1):
            out.append((f"transformer.embeddings.hybrid_model.body.block{bi + 1}.unit{u}", cin if u == 1 else cout, cout,
                        cmid, stride if u == 1 else 1))
    return out


def state_dict_schema(img_size=224, num_classes=3):
    grid = img_size // 16
    E = "transformer.embeddings"
    out = [(f"{E}.position_embeddings", (1, grid * grid, HIDDEN)),
           (f"{E}.hybrid_model.root.conv.weight", (64, 3, 7, 7)), (f"{E}.hybrid_model.root.gn.weight", (64,)),
           (f"{E}.hybrid_model.root.gn.bias", (64,))]
    for p, cin, cout, cmid, stride in _units():
        out += [(f"{p}.gn1.weight", (cmid,)), (f"{p}.gn1.bias", (cmid,)), (f"{p}.conv1.weight", (cmid, cin, 1, 1)),
                (f"{p}.gn2.weight", (cmid,)), (f"{p}.gn2.bias", (cmid,)), (f"{p}.conv2.weight", (cmid, cmid, 3, 3)),
                (f"{p}.gn3.weight", (cout,)), (f"{p}.gn3.bias", (cout,)), (f"{p}.conv3.weight", (cout, cmid, 1, 1))]
        if stride != 1 or cin != cout:
            out += [(f"{p}.downsample.weight", (cout, cin, 1, 1)), (f"{p}.gn_proj.weight", (cout,)),
                    (f"{p}.gn_proj.bias", (cout,))]
    out += [(f"{E}.patch_embeddings.weight", (HIDDEN, 1024, 1, 1)), (f"{E}.patch_embeddings.bias", (HIDDEN,))]
    for i in range(LAYERS):
        L = f"transformer.encoder.layer.{i}"
        out += [(f"{L}.attention_norm.weight", (HIDDEN,)), (f"{L}.attention_norm.bias", (HIDDEN,)),
                (f"{L}.ffn_norm.weight", (HIDDEN,)), (f"{L}.ffn_norm.bias", (HIDDEN,)),
                (f"{L}.ffn.fc1.weight", (MLP_DIM, HIDDEN)), (f"{L}.ffn.fc1.bias", (MLP_DIM,)),
                (f"{L}.ffn.fc2.weight", (HIDDEN, MLP_DIM)), (f"{L}.ffn.fc2.bias", (HIDDEN,))]
        for n in ("query", "key", "value", "out"):
            out += [(f"{L}.attn.{n}.weight", (HIDDEN, HIDDEN)), (f"{L}.attn.{n}.bias", (HIDDEN,))]
    out += [("transformer.encoder.encoder_norm.weight", (HIDDEN,)), ("transformer.encoder.encoder_norm.bias", (HIDDEN,))]

    def conv_bn(prefix, cin, cout):
        return [(f"{prefix}.0.weight", (cout, cin, 3, 3)), (f"{prefix}.1.weight", (cout,)), (f"{prefix}.1.bias", (cout,)),
                (f"{prefix}.1.running_mean", (cout,)), (f"{prefix}.1.running_var", (cout,)),
                (f"{prefix}.1.num_batches_tracked", ())]

    out += conv_bn("decoder.conv_more", HIDDEN, 512)
    cin = 512
    for i, (cout, skip) in enumerate(zip(DECODER, SKIPS)):
        out += conv_bn(f"decoder.blocks.{i}.conv1", cin + skip, cout) + conv_bn(f"decoder.blocks.{i}.conv2", cout, cout)
        cin = cout
    out += [("segmentation_head.0.weight", (num_classes, DECODER[-1], 3, 3)), ("segmentation_head.0.bias", (num_classes,))]
    return out


def seeded_state_dict(img_size=224, num_classes=3, seed=0):
    rng = np.random.RandomState(seed)
    sd = {}
    for name, shape in state_dict_schema(img_size, num_classes):
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        elif name.endswith("position_embeddings"):
            sd[name] = (0.02 * torch.from_numpy(rng.standard_normal(shape))).float()
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = (torch.from_numpy(rng.standard_normal(shape)) * math.sqrt(2.0 / fan_in)).float()
        elif len(shape) == 2:
            sd[name] = (torch.from_numpy(rng.standard_normal(shape)) * math.sqrt(1.0 / shape[1])).float()
        elif name.endswith(".weight"):  # norm scales
            sd[name] = (1 + 0.1 * torch.from_numpy(rng.standard_normal(shape))).float()
        else:
            sd[name] = (0.05 * torch.from_numpy(rng.standard_normal(shape))).float()
    return sd


def _std_conv(x, w, stride=1, padding=0):
    v, m = torch.var_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
    return F.conv2d(x, (w - m) / torch.sqrt(v + 1e-5), None, stride, padding)


def forward(sd, x, training=True):
    """logits [B, C, S, S]; also returns the batch-norm running-stat updates (training mode)."""
    new_stats = {}
    if x.shape[1] == 1:
        x = x.repeat(1, 3, 1, 1)
    b, _, in_size, _ = x.shape
    R = "transformer.embeddings.hybrid_model"
    x = _std_conv(x, sd[f"{R}.root.conv.weight"], 2, 3)
    x = F.relu(F.group_norm(x, 32, sd[f"{R}.root.gn.weight"], sd[f"{R}.root.gn.bias"], 1e-6))
    feats = [x]
    x = F.max_pool2d(x, 3, 2, 0)
    units = _units()
    for p, cin, cout, cmid, stride in units:
        res = x
        if f"{p}.downsample.weight" in sd:
            res = _std_conv(x, sd[f"{p}.downsample.weight"], stride)
            res = F.group_norm(res, cout, sd[f"{p}.gn_proj.weight"], sd[f"{p}.gn_proj.bias"], 1e-5)
        y = F.relu(F.group_norm(_std_conv(x, sd[f"{p}.conv1.weight"]), 32, sd[f"{p}.gn1.weight"], sd[f"{p}.gn1.bias"], 1e-6))
        y = F.relu(F.group_norm(_std_conv(y, sd[f"{p}.conv2.weight"], stride, 1), 32, sd[f"{p}.gn2.weight"],
                                sd[f"{p}.gn2.bias"], 1e-6))
        y = F.group_norm(_std_conv(y, sd[f"{p}.conv3.weight"]), 32, sd[f"{p}.gn3.weight"], sd[f"{p}.gn3.bias"], 1e-6)
        x = F.relu(res + y)
        block, unit = int(p.split("block")[1][0]), int(p.split("unit")[1])
        if block < 3 and unit == RESNET_UNITS[block - 1]:  # end of block1 / block2: a skip feature
            right = int(in_size / 4 / block)
            feat = x
            if x.shape[2] != right:
                pad = right - x.shape[2]
                assert 0 < pad < 3
                feat = F.pad(x, (0, pad, 0, pad))
            feats.append(feat)
    feats = feats[::-1]
    E = "transformer.embeddings"
    t = F.conv2d(x, sd[f"{E}.patch_embeddings.weight"], sd[f"{E}.patch_embeddings.bias"])
    t = t.flatten(2).transpose(-1, -2) + sd[f"{E}.position_embeddings"]
    dh = HIDDEN // HEADS
    for i in range(LAYERS):
        L = f"transformer.encoder.layer.{i}"
        h = F.layer_norm(t, (HIDDEN,), sd[f"{L}.attention_norm.weight"], sd[f"{L}.attention_norm.bias"], 1e-6)
        q, k, v = (F.linear(h, sd[f"{L}.attn.{n}.weight"], sd[f"{L}.attn.{n}.bias"]).view(b, -1, HEADS, dh).permute(0, 2, 1, 3)
                   for n in ("query", "key", "value"))
        probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, -1, HIDDEN)
        t = t + F.linear(ctx, sd[f"{L}.attn.out.weight"], sd[f"{L}.attn.out.bias"])
        h = F.layer_norm(t, (HIDDEN,), sd[f"{L}.ffn_norm.weight"], sd[f"{L}.ffn_norm.bias"], 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[f"{L}.ffn.fc1.weight"], sd[f"{L}.ffn.fc1.bias"])), sd[f"{L}.ffn.fc2.weight"],
                     sd[f"{L}.ffn.fc2.bias"])
        t = t + h
    t = F.layer_norm(t, (HIDDEN,), sd["transformer.encoder.encoder_norm.weight"], sd["transformer.encoder.encoder_norm.bias"], 1e-6)
    g = int(math.sqrt(t.shape[1]))
    x = t.permute(0, 2, 1).contiguous().view(b, HIDDEN, g, g)

    def conv_bn_relu(x, prefix):
        x = F.conv2d(x, sd[f"{prefix}.0.weight"], padding=1)
        rm, rv = sd[f"{prefix}.1.running_mean"].clone(), sd[f"{prefix}.1.running_var"].clone()
        x = F.batch_norm(x, rm, rv, sd[f"{prefix}.1.weight"], sd[f"{prefix}.1.bias"], training, 0.1, 1e-5)
        new_stats[f"{prefix}.1.running_mean"], new_stats[f"{prefix}.1.running_var"] = rm, rv
        return F.relu(x)

    x = conv_bn_relu(x, "decoder.conv_more")
    for i in range(4):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        if i < 3:
            x = torch.cat([x, feats[i]], dim=1)
        x = conv_bn_relu(conv_bn_relu(x, f"decoder.blocks.{i}.conv1"), f"decoder.blocks.{i}.conv2")
    return F.conv2d(x, sd["segmentation_head.0.weight"], sd["segmentation_head.0.bias"], padding=1), new_stats


def dice_loss(logits, target, n_classes):
    p = torch.softmax(logits, dim=1)
    loss = 0.0
    for i in range(n_classes):
        t = (target == i).float()
        inter, y_sum, z_sum = torch.sum(p[:, i] * t), torch.sum(t * t), torch.sum(p[:, i] * p[:, i])
        loss = loss + (1 - (2 * inter + 1e-5) / (z_sum + y_sum + 1e-5))
    return loss / n_classes


def train_step(sd, momentum_buffers, batch, num_classes=3, lr=0.01, momentum=0.9, weight_decay=1e-4):
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    logits, new_stats = forward(work, batch["images"])
    gt = batch["segmented"].squeeze(1)
    ce = F.cross_entropy(logits, gt.long())
    dice = dice_loss(logits, gt, num_classes)
    loss = 0.5 * ce + 0.5 * dice
    grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])))
    with torch.no_grad():
        for k in names:
            d = grads[k] + weight_decay * sd[k]
            buf = momentum_buffers.get(k)
            buf = d if buf is None else buf * momentum + d
            momentum_buffers[k] = buf
            sd[k] = sd[k] - lr * buf
        for k, v in new_stats.items():
            sd[k] = v.detach()
    return loss.detach(), ce.detach(), dice.detach(), grads, logits.detach()
