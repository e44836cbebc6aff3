"""One training iteration per segmentation model family.

Mirrors updater/segmentation_updater.py:42-106 of the reference: same class names, constructor keywords
(``em_mom`` / ``num_classes`` + the Updater dicts), batch contract (``batch['images']`` f32 [B,3,S,S] in
[-1,1], ``batch['segmented']`` int64 [B,1,S,S]) and the same order of operations inside ``update_core``:

  EMANet     forward -> EM bases moving average (no grad) -> loss.mean() -> zero_grad -> backward -> step
  TransUNet  zero_grad -> forward -> 0.5 * CE + 0.5 * Dice(softmax) -> backward -> step

MI355X specifics: the bases update is one HIP kernel (``sis_ema_update``), the EMANet loss tail is fused inside
the network (networks/ema_net/network.py), the optimizer step is the one-launch ``FusedSGD`` the builders
create, gradients are averaged over the ranks bucket by bucket over RCCL while backward is still running
(training/grad_exchange.py, or torch's DistributedDataParallel).  The whole iteration -- collectives included -- is replayed
as a hipGraph after two eager iterations (training/graph_step.py; keyword ``hip_graph=False`` or SIS_STEP_GRAPH=0 keeps it
eager, as does the DistributedDataParallel flavour).
"""
import os

import torch
from torch import nn

import sis_hip
from networks.trans_u_net.utils import DiceLoss
from training.grad_exchange import BucketedDataParallel
from training.graph_step import StepGraph
from training.loop import GradientApplier, Updater, get_current_reporter


_FUSED_LOSS = os.environ.get('SIS_FUSED_LOSS', '1') != '0'


def _unwrap(network):
    """The bare module behind a DistributedDataParallel wrapper (reference: try/except AttributeError, :58-66)."""
    return network.module if hasattr(network, 'module') and not hasattr(network, 'emau') else network


def _graphable(network, optimizer, device):
    """Whole-iteration capture needs a HIP device and the fused optimizer; under data parallelism also an exchange whose
    collectives are stream work (training/grad_exchange.py over RCCL) -- torch's DistributedDataParallel reducer stays eager."""
    if isinstance(network, nn.parallel.DistributedDataParallel) or not hasattr(optimizer, 'push_hyper'):
        return False
    if isinstance(network, BucketedDataParallel) and not network.capturable():
        return False
    return torch.device(device if not isinstance(device, int) else f'cuda:{device}').type == 'cuda'


class _GraphedUpdater(Updater):
    """``update_core`` = next batch -> ``_iteration(batch)`` (eager or as a replayed hipGraph) -> report."""
    report_prefix = 'loss'

    def __init__(self, *args, **kwargs):
        hip_graph = kwargs.pop('hip_graph', True)
        super().__init__(*args, **kwargs)
        self._step_graph = StepGraph(warmup=2, enabled=bool(hip_graph) and _graphable(
            self.networks['segmentation'], self.optimizers['main'], self.device))

    def _iteration(self, batch):
        raise NotImplementedError

    def update_core(self):
        batch = self.next_batch('images')  # restarts a finite loader at the epoch boundary
        batch = {key: value.to(self.device, non_blocking=True) for key, value in batch.items()}
        observed = self._step_graph.run(batch, self._iteration, [self.optimizers['main']])
        get_current_reporter().add_observation(observed, self.report_prefix)


class EMANetUpdater(_GraphedUpdater):
    def __init__(self, *args, **kwargs):
        self.em_mom = kwargs.pop('em_mom')
        super().__init__(*args, **kwargs)

    def _iteration(self, batch):
        network = self.networks['segmentation']
        optimizer = self.optimizers['main']

        loss, mu = network(batch['images'], torch.squeeze(batch['segmented'], dim=1))

        with torch.no_grad():
            bases = _unwrap(network).emau.mu
            if bases.is_cuda:
                sis_hip.ema_update(bases, mu, self.em_mom)
            else:
                raise RuntimeError("mu must be a CUDA tensor")

        loss = loss.mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        return {'softmax': loss.detach()}


class _CeDiceFn(torch.autograd.Function):
    """0.5 * CrossEntropy + 0.5 * Dice(softmax) on the logits as the segmentation head wrote them (bf16 under autocast), one
    pass over them per direction (csrc/loss_ops.hip) -> (combined, CE, Dice); only ``combined`` carries a gradient."""

    @staticmethod
    def forward(ctx, logits, labels):
        out, stats = sis_hip.ce_dice_fwd(logits, labels)
        ctx.save_for_backward(logits, labels, stats)
        ctx.mark_non_differentiable(out)
        combined = out[0].clone()
        return combined, out

    @staticmethod
    def backward(ctx, grad_combined, _grad_out):
        logits, labels, stats = ctx.saved_tensors
        return sis_hip.ce_dice_bwd(grad_combined, logits, labels, stats), None


class TransUNetUpdater(_GraphedUpdater):
    def __init__(self, *args, **kwargs):
        num_classes = kwargs.pop('num_classes')
        amp = kwargs.pop('amp', None)  # None / 'bf16': the reference is fp32; bf16 autocast is this build's option
        super().__init__(*args, **kwargs)
        self.ce_loss = nn.CrossEntropyLoss()
        self.dice_loss = DiceLoss(num_classes)
        self.amp_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(amp)

    def _iteration(self, batch):
        network = self.networks['segmentation']

        with GradientApplier([network], [self.optimizers['main']]):
            # bf16 autocast covers the network only (GEMMs / convolutions / attention on the bf16 matrix cores,
            # fp32 master weights, fp32 SGD); the losses are evaluated on fp32 logits as in the reference.
            with torch.autocast(device_type='cuda', dtype=self.amp_dtype or torch.bfloat16,
                                enabled=self.amp_dtype is not None):
                prediction = network(batch['images'])
            ground_truth = torch.squeeze(batch['segmented'], dim=1).long()
            if _FUSED_LOSS and sis_hip.ce_dice_supported(prediction, ground_truth) and prediction.shape[1] == self.dice_loss.n_classes:
                loss, parts = _CeDiceFn.apply(prediction, ground_truth)   # the logits stay in the head's dtype
                loss_ce, loss_dice = parts[1], parts[2]
            else:
                prediction = prediction.float()
                loss_ce = self.ce_loss(prediction, ground_truth)
                loss_dice = self.dice_loss(prediction, ground_truth, softmax=True)
                loss = 0.5 * loss_ce + 0.5 * loss_dice
            loss.backward()

        return {'combined': loss.detach(), 'CE': loss_ce.detach(), 'Dice': loss_dice.detach()}
