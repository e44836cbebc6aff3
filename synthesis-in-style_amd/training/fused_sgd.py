"""``torch.optim.SGD`` semantics on one HIP launch per step (``sis_sgd_momentum``).

Same constructor arguments and ``param_groups`` layout as ``torch.optim.SGD(params, lr, momentum,
weight_decay)`` (what training_builder/ema_net_train_builder.py:27-48 and trans_u_net_train_builder.py:39-40
build), so LR schedulers and checkpoints (``state_dict`` with ``momentum_buffer`` per parameter) keep working.
Per step it launches ONE kernel over a device-resident table of (param, grad, momentum buffer) chunks instead
of a foreach sequence.  Parameter and momentum pointers are fixed; the gradient column is refreshed on the host
with numpy whenever gradient storage moved (one 20 KB pinned, non-blocking copy) -- which makes
``zero_grad(set_to_none=True)`` the cheap default: no 178 fill kernels, and autograd assigns fresh gradients instead
of launching 178 accumulate-adds into stale ones (measured 1.8 ms of a 45 ms EMANet-50 step).  With
DistributedDataParallel(gradient_as_bucket_view=True) gradients live in the buckets and never move.

hipGraph capture (training/graph_step.py): when ``step()`` runs on a capturing stream it launches the variant
that reads lr / weight decay / momentum from a 9-float device tensor (``sis_sgd_momentum_dev``), so the captured
launch holds nothing but pointers; ``push_hyper()`` refreshes that tensor from ``param_groups`` before every replay
and the per-iteration LR schedule keeps working.
"""
import numpy as np
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
        self._grad_key = None
        self._steps = 0
        self._hyper = None        # device float32[9]: lr[4], wd[4], momentum (capturable step)
        self._hyper_hosts = None  # pinned staging ring + the events guarding its reuse
        self._capture_host = None  # pinned source of the pointer-table copy recorded in a captured step
        self._shadows = {}         # parameter -> bfloat16 copy this optimizer keeps current (register_shadow)

    def register_shadow(self, param, shadow):
        """``shadow``: a contiguous bfloat16 tensor (may be a slice of a larger buffer) with ``param``'s element count and
        current value.  Every step rewrites it with the rounded new parameter in the SAME launch that updates the fp32
        master weight, so bf16 layers (TransUNet's encoder Linear layers under autocast) never launch a cast."""
        if shadow.dtype != torch.bfloat16 or not shadow.is_contiguous() or shadow.numel() != param.numel():
            raise ValueError("shadow must be a contiguous bfloat16 tensor of the parameter's size")
        self._shadows[param] = shadow
        self._grad_key = None  # the pointer table gains a column entry

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)

    def _layout(self, entries):
        """Static part of the table: which chunk belongs to which tensor at which offset."""
        chunk = sis_hip.sgd_chunk_elems()
        owner, offset, count, group = [], [], [], []
        for ti, (gi, p, _, _) in enumerate(entries):
            n = p.numel()
            for off in range(0, n, chunk):
                owner.append(ti)
                offset.append(4 * off)
                count.append(min(chunk, n - off) | (gi << 48))
        self._owner = np.asarray(owner, dtype=np.int64)
        self._offset = np.asarray(offset, dtype=np.int64)
        self._count = np.asarray(count, dtype=np.int64)
        self._n_chunks = len(owner)
        # pinned staging ring: earlier steps' non-blocking copies may still be in flight, so every buffer carries the
        # event recorded behind its last copy and is only rewritten after that event (as push_hyper() does)
        device = entries[0][1].device
        self._on_gpu = device.type == 'cuda'  # (host-only runs exist for the gloo rehearsal of the bucket aliasing logic)
        self._hosts = [[self._staging((self._n_chunks, 5)), None] for _ in range(4)]
        self._flip = 0
        self._table = torch.empty((self._n_chunks, 5), dtype=torch.int64, device=device)

    def _staging(self, shape):
        host = torch.empty(shape, dtype=torch.int64)
        return host.pin_memory() if self._on_gpu else host

    def push_hyper(self):
        """Copies the current lr / weight decay / momentum of ``param_groups`` into the device tensor the captured
        step reads (non-blocking, from a ring of pinned buffers each guarded by an event)."""
        device = self.param_groups[0]['params'][0].device
        if self._hyper is None:
            self._hyper = torch.zeros(9, dtype=torch.float32, device=device)
            self._hyper_hosts = [[torch.zeros(9, dtype=torch.float32).pin_memory(), None] for _ in range(4)]
            self._hyper_slot = 0
        if self._table is not None and (self._capture_host is None or self._capture_host.shape[0] != self._n_chunks):
            self._capture_host = torch.empty((self._n_chunks, 5), dtype=torch.int64).pin_memory()
        self._hyper_slot = (self._hyper_slot + 1) % len(self._hyper_hosts)
        slot = self._hyper_hosts[self._hyper_slot]
        if slot[1] is not None:
            slot[1].synchronize()  # only blocks when the host is a whole ring ahead of the device
        host = slot[0].numpy()
        host[:] = 0.0
        for gi, group in enumerate(self.param_groups):
            host[gi], host[4 + gi] = group['lr'], group['weight_decay']
        host[8] = self.param_groups[0]['momentum']
        self._hyper.copy_(slot[0], non_blocking=True)
        slot[1] = torch.cuda.current_stream(device).record_event()

    def _upload(self, entries, capturing=False):
        ptrs = np.asarray([(p.data_ptr(), g.data_ptr(), b.data_ptr()) for _, p, g, b in entries], dtype=np.int64)
        shadow = np.asarray([self._shadows[p].data_ptr() if p in self._shadows else 0 for _, p, _, _ in entries], dtype=np.int64)
        if capturing:
            # the captured copy node re-reads its source on every replay: it gets a buffer nothing else writes
            # (allocated by push_hyper(): pinning memory is not allowed while a stream is capturing)
            pinned, slot = self._capture_host, None
        else:
            self._flip = (self._flip + 1) % len(self._hosts)
            slot = self._hosts[self._flip]
            if slot[1] is not None:
                slot[1].synchronize()  # only blocks when the host is a whole ring ahead of the device
            pinned = slot[0]
        host = pinned.numpy()
        host[:, :3] = ptrs[self._owner] + self._offset[:, None]
        host[:, 3] = self._count
        sh = shadow[self._owner]
        host[:, 4] = np.where(sh != 0, sh + self._offset // 2, 0)  # bf16 elements: half the byte offset of the fp32 chunk
        self._table.copy_(pinned, non_blocking=True)
        if slot is not None and self._on_gpu:
            slot[1] = torch.cuda.current_stream(self._table.device).record_event()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        sis_hip.flush_deferred()   # (normally drained at the end of the backward: a no-op then)
        entries, fresh = [], False
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue  # torch.optim.SGD skips parameters without a gradient (e.g. EMANet's emau.conv1)
                sis_hip.require_device(p, "parameter")
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedSGD needs contiguous float32 parameters and gradients")
                state = self.state[p]
                if state.get('momentum_buffer') is None:
                    state['momentum_buffer'] = torch.empty_like(p)
                    state['fresh'] = True
                fresh = fresh or state.get('fresh', False)
                entries.append((gi, p, p.grad, state['momentum_buffer']))
        if not entries:
            return loss
        if fresh and not all(self.state[p].get('fresh', False) for _, p, _, _ in entries):
            raise RuntimeError("FusedSGD: parameters gained gradients after the first step; rebuild the optimizer")
        static_key = tuple(id(p) for _, p, _, _ in entries)
        if static_key != self._table_key:
            self._layout(entries)
            self._table_key = static_key
            self._grad_key = None
        capturing = self._on_gpu and torch.cuda.is_current_stream_capturing()
        if capturing and (fresh or self._hyper is None or self._capture_host is None):
            raise RuntimeError("FusedSGD: capture needs one eager step() and a push_hyper() call first")
        for _, p, _, _ in entries:
            # the update below bypasses torch's version counter: modules caching derived tensors (bf16 shadows that are
            # NOT registered here) watch this counter next to ``param._version``
            p._sis_raw_updates = getattr(p, '_sis_raw_updates', 0) + 1
        grad_key = tuple(g.data_ptr() for _, _, g, _ in entries) + tuple(p.data_ptr() for _, p, _, _ in entries)
        if grad_key != self._grad_key or capturing:
            self._upload(entries, capturing)
            self._grad_key = None if capturing else grad_key
        if capturing:
            sis_hip.sgd_momentum_dev(self._table, self._n_chunks, self._hyper)
            self._steps += 1
            return loss
        lrs = [g['lr'] for g in self.param_groups]
        wds = [g['weight_decay'] for g in self.param_groups]
        sis_hip.sgd_momentum(self._table, self._n_chunks, lrs, wds, self.param_groups[0]['momentum'], fresh)
        if fresh:
            for _, p, _, _ in entries:
                self.state[p]['fresh'] = False
        self._steps += 1
        return loss
