"""``torch.optim.SGD`` semantics on one HIP launch per step (``sis_sgd_momentum``).

Same constructor arguments and ``param_groups`` layout as ``torch.optim.SGD(params, lr, momentum,
weight_decay)`` (what training_builder/ema_net_train_builder.py:27-48 and trans_u_net_train_builder.py:39-40
build), so LR schedulers and checkpoints (``state_dict`` with ``momentum_buffer`` per parameter) keep working.
Per step it launches ONE kernel over a device-resident table of (param, grad, momentum buffer) chunks instead
of a foreach sequence; the table is rebuilt only when a parameter or gradient storage moves.

``zero_grad`` defaults to ``set_to_none=False`` -- the behaviour of the torch 1.9 the reference pins
(requirements.txt:1) -- which also keeps gradient storage, and therefore the table, stable.
"""
import torch
from torch.optim.optimizer import Optimizer

import sis_hip


class FusedSGD(Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if momentum < 0 or lr < 0 or weight_decay < 0:
            raise ValueError("lr, momentum and weight_decay must be non-negative")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        if len(self.param_groups) > 4:
            raise ValueError("FusedSGD supports at most 4 parameter groups")
        momenta = {g['momentum'] for g in self.param_groups}
        if len(momenta) != 1:
            raise ValueError("all parameter groups must share one momentum")
        self._table = None
        self._table_key = None
        self._steps = 0

    def zero_grad(self, set_to_none: bool = False):
        super().zero_grad(set_to_none=set_to_none)

    def _build_table(self, entries, device):
        chunk = sis_hip.sgd_chunk_elems()
        rows = []
        for gi, p, g, buf in entries:
            n = p.numel()
            for off in range(0, n, chunk):
                cnt = min(chunk, n - off)
                rows.append((p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, buf.data_ptr() + 4 * off, cnt | (gi << 48)))
        self._table = torch.tensor(rows, dtype=torch.int64).to(device)
        self._n_chunks = len(rows)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries, key, fresh = [], [], False
        device = None
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue  # torch.optim.SGD skips parameters without a gradient (e.g. EMANet's emau.conv1)
                sis_hip.require_device(p, "parameter")
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedSGD needs contiguous float32 parameters and gradients")
                state = self.state[p]
                if 'momentum_buffer' not in state or state['momentum_buffer'] is None:
                    state['momentum_buffer'] = torch.empty_like(p)
                    state['fresh'] = True
                fresh = fresh or state.get('fresh', False)
                entries.append((gi, p, p.grad, state['momentum_buffer']))
                key.append((p.data_ptr(), p.grad.data_ptr(), state['momentum_buffer'].data_ptr()))
                device = p.device
        if not entries:
            return loss
        if fresh and not all(self.state[p].get('fresh', False) for _, p, _, _ in entries):
            raise RuntimeError("FusedSGD: parameters gained gradients after the first step; rebuild the optimizer")
        key = tuple(key)
        if key != self._table_key:
            self._build_table(entries, device)
            self._table_key = key
        lrs = [g['lr'] for g in self.param_groups]
        wds = [g['weight_decay'] for g in self.param_groups]
        sis_hip.sgd_momentum(self._table, self._n_chunks, lrs, wds, self.param_groups[0]['momentum'], fresh)
        if fresh:
            for _, p, _, _ in entries:
                self.state[p]['fresh'] = False
        self._steps += 1
        return loss
