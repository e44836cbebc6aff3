"""EMANet (dilated ResNet backbone + Expectation-Maximisation Attention Unit) for MI355X.

Drop-in for /root/reference/stylegan_code_finder/networks/ema_net/network.py: same constructor, same
``forward(img, lbl=None, size=None)`` contract (``(loss[B], mu)`` when training with labels, logits otherwise),
same ``state_dict`` keys (353 for ResNet-50: ``extractor.{0..7}...``, ``fc0``, ``emau``, ``fc1``, ``fc2``;
checked against the reference's key list in the test-suite), same initialisation statistics.

Differences in execution:
* the loss tail -- bilinear upsampling of the stride-8 logits to label resolution, log-softmax, NLL with
  ``ignore_label`` and the per-sample spatial mean (reference :305-311, :319-327) -- is ONE hand-written HIP
  kernel each way (``sis_upsample_ce_fwd/bwd``); the full-resolution logits are never materialised;
* "SynchronizedBatchNorm2d" is what it effectively is under the reference's DistributedDataParallel launch:
  per-GPU ``F.batch_norm`` with momentum 3e-4 (bn_lib/nn/modules/batchnorm.py:51-56); the DataParallel-era
  synchronisation machinery is not carried over (never active under train.py);
* 3x3 (Winograd forward / data gradient / weight gradient), 1x1 and stride-2 convolutions, batch norm (+ ReLU, + residual),
  max pooling and the EM iterations run on the hand-written kernels of csrc/ (``networks/hip_conv.py``, ``sis_bn_*``,
  ``sis_max_pool2d``, ``sis_ema_*``); what stays on the ROCm libraries through ATen is the 3-channel stem convolution, the
  dilation-16 layers whose 2x2 sub-images the tile plan rejects, and the EMAU's small bmm (0.9 of 29.9 ms per step,
  profiles/r02_z_emanet_step_breakdown.txt).
"""
import math
import contextlib
import os
import pathlib
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.nn.modules.batchnorm import _BatchNorm

import sis_hip
from networks.base_segmenter import BaseSegmenter
from networks.hip_conv import HipConv2d, _batch_to_space, _space_to_batch, conv3x3, pointwise_with_skip
from networks.hip_pool import HipMaxPool2d

BN_MOM = 3e-4
_HIP_EMAU = os.environ.get('SIS_HIP_EMAU', '1') != '0'   # 0: the EM rounds as torch.bmm / softmax launches (A/B runs)
_RELU_MASK = os.environ.get('SIS_BN_RELU_MASK', '1') != '0'
RESNET_BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class _FusedBatchNormAct(Function):
    """y = [relu]( batch_norm(x) [+ residual] ) on the HIP kernels of csrc/bn_ops.hip (one statistics pass and one
    apply pass forward; one reduction pass and one apply pass backward, residual gradient included)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu):
        if training and sis_hip.bn_fused_supported(x):
            # a channel's batch * H * W values fit one workgroup (the 32 x 32 layers): statistics and apply in one launch, x read once
            y, mean, invstd, mask = sis_hip.bn_fused_fwd(x, residual, weight, bias, running_mean, running_var, eps, momentum, relu,
                                                         want_mask=relu and _RELU_MASK)
            ctx.save_for_backward(x, None if mask is not None else y, mean, invstd, weight, mask)
            ctx.relu, ctx.training, ctx.has_residual = relu, training, residual is not None
            return y
        if training:
            mean, invstd = sis_hip.bn_stats(x, running_mean, running_var, eps, momentum)
        else:
            mean, invstd = running_mean, torch.rsqrt(running_var + eps)
        # with a ReLU the backward only needs the SIGN of y: the apply pass leaves one bit per element (csrc/bn_ops.hip) and
        # both backward passes read that instead of the fp32 tensor
        mask = None
        if relu and training and _RELU_MASK:
            y, mask = sis_hip.bn_act_fwd(x, residual, mean, invstd, weight, bias, relu, want_mask=True)
        else:
            y = sis_hip.bn_act_fwd(x, residual, mean, invstd, weight, bias, relu)
        ctx.save_for_backward(x, None if mask is not None else y, mean, invstd, weight, mask)
        ctx.relu, ctx.training, ctx.has_residual = relu, training, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("fused batch norm: backward through evaluation-mode statistics is not implemented")
        x, y, mean, invstd, weight, mask = ctx.saved_tensors
        need_res = ctx.has_residual and ctx.needs_input_grad[1]
        dx, dres, dgamma, dbeta = sis_hip.bn_act_bwd(dy.contiguous(), y, x, mean, invstd, weight, ctx.relu, need_res, mask=mask)
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    """Per-process batch norm (what the reference's class amounts to under DistributedDataParallel); always batch /
    running statistics through the functional path, so ``num_batches_tracked`` stays untouched as in the reference.
    ``forward(x, residual=None, relu=False)`` additionally fuses the residual add and ReLU that follow it in the
    bottlenecks / ConvBNReLU blocks; on a HIP device with fp32 NCHW input the whole thing runs on the hand-written
    kernels, otherwise (other dtypes / layouts) on the ATen composite."""

    def forward(self, input, residual=None, relu=False):
        fused = (sis_hip.bn_supported(input) and self.weight is not None and self.bias is not None
                 and (residual is None or (sis_hip.bn_supported(residual) and residual.shape == input.shape))
                 and (self.training or not torch.is_grad_enabled()))
        if fused:
            return _FusedBatchNormAct.apply(input, residual, self.weight, self.bias, self.running_mean, self.running_var,
                                            self.training, self.momentum, self.eps, relu)
        out = F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                           self.momentum, self.eps)
        if residual is not None:
            out = out + residual
        return F.relu(out, inplace=True) if relu else out


norm_layer = partial(SynchronizedBatchNorm2d, momentum=BN_MOM)


def _init_weights(module):
    """He-normal on fan-out for convolutions, unit scale / zero shift for norms (reference :88-98)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / fan_out))
        elif isinstance(m, _BatchNorm):
            m.weight.data.fill_(1)
            if m.bias is not None:
                m.bias.data.zero_()


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, previous_dilation=1):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = HipConv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = HipConv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.dilation = dilation
        self.stride = stride

    def forward(self, x, sub_images=False):
        """``sub_images``: x is the space-to-batch arrangement [B d^2, C, H/d, W/d] of the unit's input for d = this unit's
        dilation (``run_units``): the dilated 3x3 layer is then a plain one, every other layer of the unit is pointwise or a
        per-channel statistic and does not see the difference."""
        fused = pointwise_with_skip(self.conv1, x) if self.downsample is None else None
        if fused is not None:
            y, shortcut = fused   # conv1(x) and the identity shortcut: their two gradients are summed in conv1's data-gradient kernel
        else:
            shortcut = x if self.downsample is None else self.downsample(x)
            y = self.conv1(x)
        y = self.bn1(y, relu=True)
        y = self.bn2(self._conv2_on_sub_images(y) if sub_images else self.conv2(y), relu=True)
        return self.bn3(self.conv3(y), residual=shortcut, relu=True)

    def _conv2_on_sub_images(self, y):
        self.conv2._takes_wino = True
        return conv3x3(y, self.conv2.weight, 1, self.conv2._wino_banked, self.conv2._defer_wgrad)

    def sub_image_eligible(self, x):
        """The unit can run on the sub-image arrangement of x: a stride-1 dilated fp32 unit whose 3x3 layer the Winograd kernel
        takes at the sub-image size."""
        d = self.dilation
        if not (_SUB_IMAGE_UNITS and d > 1 and self.stride == 1 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and not torch.is_autocast_enabled() and x.shape[2] % d == 0 and x.shape[3] % d == 0 and self.conv2.bias is None):
            return False
        b, _, h, w = x.shape
        probe = x.new_empty((b * d * d, self.conv2.in_channels, h // d, w // d))
        return bool(sis_hip.conv3x3_supported(probe, self.conv2.weight, 1))


# Consecutive dilated units with the SAME dilation d (EMANet-50 at output stride 8: layer3's units 2-6 and layer4's first, d = 2)
# run on the space-to-batch arrangement of their input: one rearrangement in, one out, instead of two around every 3x3 layer in
# each direction (hip_conv._Conv3x3Function: 4 strided copies per dilated layer and iteration, 24 -> 4 for that run of six units;
# SIS_SUB_IMAGE_UNITS=0: every unit on the plain arrangement, A/B runs).
_SUB_IMAGE_UNITS = os.environ.get('SIS_SUB_IMAGE_UNITS', '1') != '0'


def run_units(units, x):
    """``units`` (Bottlenecks of one or several stages, in order) applied to x, runs of equal dilation on sub-images."""
    units = list(units)
    i = 0
    while i < len(units):
        unit = units[i]
        d = getattr(unit, 'dilation', 1)
        if d > 1 and unit.sub_image_eligible(x):
            j = i
            while j < len(units) and getattr(units[j], 'dilation', 1) == d and units[j].sub_image_eligible(x):
                j += 1
            if j - i >= 2:   # (a single unit gains nothing over the rearrangement inside its 3x3 layer)
                xs = _space_to_batch(x, d).contiguous()
                for u in units[i:j]:
                    xs = u(xs, sub_images=True)
                x = _batch_to_space(xs, d).contiguous()
                i = j
                continue
        x = unit(x)
        i += 1
    return x


class ResNet(nn.Module):
    """Deep-stem ResNet with output stride 8 or 16 (dilated layer3 / layer4, multi-grid [1,2,4] in layer4)."""

    def __init__(self, block, layers, num_classes=1000, stride=8):
        super().__init__()
        self.inplanes = 128
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, 64, 3, 2, 1, bias=False), norm_layer(64), nn.ReLU(inplace=True),
            HipConv2d(64, 64, 3, 1, 1, bias=False), norm_layer(64), nn.ReLU(inplace=True),
            HipConv2d(64, 128, 3, 1, 1, bias=False))
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        if stride == 16:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=2, grids=[1, 2, 4])
        elif stride == 8:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4, grids=[1, 2, 4])
        else:
            raise RuntimeError(f'=> unsupported output stride: {stride}')
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        _init_weights(self)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, grids=None):
        grids = grids or [1] * blocks
        if dilation not in (1, 2, 4):
            raise RuntimeError(f'=> unknown dilation size: {dilation}')
        out_planes = planes * block.expansion
        downsample = None
        if stride != 1 or self.inplanes != out_planes:
            downsample = nn.Sequential(HipConv2d(self.inplanes, out_planes, 1, stride, bias=False),
                                       norm_layer(out_planes))
        first_dilation = 2 if dilation == 4 else 1
        units = [block(self.inplanes, planes, stride, dilation=first_dilation, downsample=downsample,
                       previous_dilation=dilation)]
        self.inplanes = out_planes
        units += [block(out_planes, planes, dilation=dilation * grids[i], previous_dilation=dilation)
                  for i in range(1, blocks)]
        return nn.Sequential(*units)

    @staticmethod
    def stem(conv1, bn1, x):
        """conv-bn-relu x3 of the deep stem with the norm / activation pairs fused."""
        x = conv1[1](conv1[0](x), relu=True)
        x = conv1[4](conv1[3](x), relu=True)
        return bn1(conv1[6](x), relu=True)

    def forward(self, x):
        x = self.maxpool(self.stem(self.conv1, self.bn1, x))
        x = run_units(list(self.layer3) + list(self.layer4), self.layer2(self.layer1(x)))
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))


def resnet(n_layers, stride, use_pretrained_resnet, pretrained_path):
    net = ResNet(Bottleneck, layers=RESNET_BLOCKS[n_layers], stride=stride)
    if use_pretrained_resnet:
        path = pathlib.Path(pretrained_path)
        assert path.exists(), f'There does not seem to be a pretrained model at {pretrained_path}. Make sure ' \
                              f'you downloaded the correct model and saved it there.'
        net.load_state_dict(torch.load(path), strict=False)
    return net


class ConvBNReLU(nn.Module):
    def __init__(self, c_in, c_out, kernel_size, stride, padding, dilation):
        super().__init__()
        self.conv = HipConv2d(c_in, c_out, kernel_size, stride, padding, dilation, bias=False)
        self.bn = norm_layer(c_out)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.bn(self.conv(x), relu=True)


class EMAU(nn.Module):
    """Expectation-Maximisation Attention Unit: ``stage_num`` E/M rounds over ``k`` bases held in the buffer
    ``mu`` [1, c, k]; the rounds run outside autograd, only the reconstruction ``mu z^T`` is differentiated
    through (and, as in the reference, only w.r.t. nothing upstream: conv1 receives no gradient)."""

    def __init__(self, c, k, stage_num=3):
        super().__init__()
        self.stage_num = stage_num
        mu = torch.empty(1, c, k).normal_(0, math.sqrt(2. / k))
        self.register_buffer('mu', self._l2norm(mu, dim=1))
        self.conv1 = HipConv2d(c, c, 1)
        self.conv2 = nn.Sequential(HipConv2d(c, c, 1, bias=False), norm_layer(c))
        _init_weights(self)

    def forward(self, x):
        idn = x
        x = self.conv1(x)
        b, c, h, w = x.size()
        if _HIP_EMAU and sis_hip.emau_supported(x, self.mu):
            # E / M rounds and the reconstruction on the fp32 matrix cores, 7 launches (csrc/emau.hip); like the reference's
            # no_grad block the result carries no graph: conv1 receives no gradient, conv2 sees a constant input
            with torch.no_grad():
                x, mu = sis_hip.emau_forward(x.detach(), self.mu, self.stage_num)
            x = self.conv2[1](self.conv2[0](x), residual=idn, relu=True)
            return x, mu
        if x.is_cuda:
            sis_hip.library_call("ema_net.EMAU (bmm / softmax rounds)")
        x = x.view(b, c, h * w)
        mu = self.mu.repeat(b, 1, 1)
        with torch.no_grad():
            x_t = x.permute(0, 2, 1)
            for _ in range(self.stage_num):
                z = F.softmax(torch.bmm(x_t, mu), dim=2)              # E: responsibilities  [b, n, k]
                z_ = z / (1e-6 + z.sum(dim=1, keepdim=True))
                mu = self._l2norm(torch.bmm(x, z_), dim=1)             # M: bases             [b, c, k]
        x = F.relu(mu.matmul(z.permute(0, 2, 1)).view(b, c, h, w), inplace=True)
        x = self.conv2[1](self.conv2[0](x), residual=idn, relu=True)
        return x, mu

    @staticmethod
    def _l2norm(inp, dim):
        return inp / (1e-6 + inp.norm(dim=dim, keepdim=True))


class _UpsampleCrossEntropy(Function):
    """loss[b] = mean_{y,x} NLL(log_softmax(bilinear_{align_corners}(logits))[b,:,y,x], labels[b,y,x])."""

    @staticmethod
    def forward(ctx, logits, labels, size, ignore_index):
        ctx.save_for_backward(logits, labels)
        ctx.size, ctx.ignore_index = size, ignore_index
        return sis_hip.upsample_ce_fwd(logits, labels, size, ignore_index)

    @staticmethod
    def backward(ctx, grad_loss):
        logits, labels = ctx.saved_tensors
        return sis_hip.upsample_ce_bwd(grad_loss.contiguous(), logits, labels, ctx.size, ctx.ignore_index), None, None, None


class CrossEntropyLoss2d(nn.Module):
    """Per-sample mean cross entropy over [B, C, H, W] logits (reference :319-327); kept for callers that hold
    full-resolution predictions.  EMANet.forward itself uses the fused upsample + loss kernel."""

    def __init__(self, weight=None, reduction='none', ignore_index=255):
        super().__init__()
        self.nll_loss = nn.NLLLoss(weight, reduction=reduction, ignore_index=ignore_index)
        self.ignore_index = ignore_index

    def forward(self, inputs, targets):
        return self.nll_loss(F.log_softmax(inputs, dim=1), targets).mean(dim=2).mean(dim=1)


_WINO_BANK = os.environ.get('SIS_WINO_BANK', '1') != '0'  # 0: one Winograd weight-transform launch per 3x3 layer and step


@contextlib.contextmanager
def _winograd_bank(model):
    """Forward / adjoint Winograd images of every 3x3 layer that takes the Winograd path (``HipConv2d._takes_wino``, set by the
    layer's first forward), written by ONE launch per step (``sis_hip.WinogradPackBank``) and handed to the layers for the
    duration of this forward."""
    layers = []
    if _WINO_BANK and torch.is_grad_enabled():
        layers = [m for m in model.modules() if isinstance(m, HipConv2d) and m._takes_wino and m.weight.is_cuda
                  and m.weight.dtype == torch.float32 and m.weight.is_contiguous() and m.kernel_size == (3, 3)]
    if layers:
        bank = getattr(model, '_wino_bank', None)
        if bank is None or len(bank.weights) != len(layers) or any(a is not m.weight for a, m in zip(bank.weights, layers)) \
                or not bank.current():
            bank = model._wino_bank = sis_hip.WinogradPackBank([m.weight for m in layers])
        bank.refresh()
        for m, u, ua in zip(layers, bank.u, bank.u_adjoint):
            m._wino_banked = (u, ua)
    try:
        yield
    finally:
        for m in layers:
            m._wino_banked = None


class EMANet(BaseSegmenter):
    def __init__(self, num_classes, n_layers, stride=8, stage_num=3, ignore_label=255, background_class_id: int = 0,
                 min_confidence: float = 0.0, min_contour_area: int = 0, use_pretrained_resnet=True,
                 pretrained_path: str = ""):
        super().__init__(background_class_id, min_confidence, min_contour_area)
        self.num_classes = num_classes
        backbone = resnet(n_layers, stride, use_pretrained_resnet, pretrained_path)
        self.extractor = nn.Sequential(backbone.conv1, backbone.bn1, backbone.relu, backbone.maxpool,
                                       backbone.layer1, backbone.layer2, backbone.layer3, backbone.layer4)
        self.fc0 = ConvBNReLU(2048, 512, 3, 1, 1, 1)
        self.emau = EMAU(512, 64, stage_num)
        self.fc1 = nn.Sequential(ConvBNReLU(512, 256, 3, 1, 1, 1), nn.Dropout2d(p=0.1))
        self.fc2 = HipConv2d(256, num_classes, 1)
        self.crit = CrossEntropyLoss2d(ignore_index=ignore_label, reduction='none')
        self.ignore_label = ignore_label
        for m in self.modules():   # every weight is used once per forward: weight gradients may be batched (hip_conv._may_defer)
            if isinstance(m, HipConv2d):
                m._defer_wgrad = True

    def features(self, img):
        ex = self.extractor  # (stem convs, bn1, relu, maxpool, layer1..4): same modules, fused norm/activation calls
        x = ex[3](ResNet.stem(ex[0], ex[1], img))
        x = ex[5](ex[4](x))
        return run_units(list(ex[6]) + list(ex[7]), x)   # layer3 + layer4: runs of equal dilation on sub-images

    def logits(self, img):
        x = self.fc0(self.features(img))
        x, mu = self.emau(x)
        return self.fc2(self.fc1(x)), mu

    def forward(self, img, lbl=None, size=None):
        with _winograd_bank(self):   # one launch transforms the weights of every 3x3 layer (19 launches per step before)
            x, mu = self.logits(img)
        if size is None:
            size = img.size()[-2:]
        if self.training and lbl is not None:
            sis_hip.require_device(x, "img")  # the loss tail is a HIP kernel: no CPU fallback
            if x.dtype == torch.float32 and x.shape[1] <= 32:
                return _UpsampleCrossEntropy.apply(x, lbl, (int(size[0]), int(size[1])), self.ignore_label), mu
            # > 32 classes or reduced precision: library composite on the device (same math, three passes)
            pred = F.interpolate(x.float(), size=size, mode='bilinear', align_corners=True)
            return self.crit(pred, lbl), mu
        return F.interpolate(x, size=size, mode='bilinear', align_corners=True)

    def predict_classes(self, x: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.forward(x), dim=1, keepdim=True)
