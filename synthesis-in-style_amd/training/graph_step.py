"""One training iteration as a hipGraph.

A segmentation step is ~1300 (EMANet-50) to ~4000 (TransUNet, bf16) kernel launches of mostly tens of microseconds:
in bf16 the host cannot issue them as fast as the device retires them (measured 23 ms of idle device per 91 ms
TransUNet step).  ``StepGraph`` runs the first ``warmup`` iterations eagerly (library algorithm selection, allocator
high-water mark, momentum buffers), then captures ONE iteration -- forward, loss, backward, fused SGD -- into a
``torch.cuda.CUDAGraph`` (hipGraph on ROCm) and replays it: per iteration the host copies the batch into the
graph's static inputs, refreshes the optimizer's device-side hyper-parameters (the LR schedule keeps running on the
host) and launches the graph.

Data parallel: the bucketed gradient exchange of training/grad_exchange.py is stream work (RCCL collectives on a side
stream forked from and joined to the step's stream), so the captured iteration contains it; its bucket plan is fixed in
the first eager iteration, the second one rehearses the bucketed path, the third is captured.  torch's
DistributedDataParallel flavour stays eager (its reducer synchronises with the host).

A capture that fails leaves the step eager with a warning and ``capture_error`` set; ``strict=True`` (bench.py) or
SIS_STEP_GRAPH_STRICT=1 re-raises instead, so that a silent regression to the eager path cannot pass as a measurement.
"""
import os
from typing import Callable, Dict, Iterable

import torch


class StepGraph:
    def __init__(self, warmup: int = 2, enabled: bool = True, strict: bool = False):
        self.warmup = max(int(warmup), 1)  # FusedSGD needs one eager step before capture
        self.requested = enabled and os.environ.get("SIS_STEP_GRAPH", "1") != "0"
        self.enabled = self.requested
        self.strict = strict or os.environ.get("SIS_STEP_GRAPH_STRICT", "0") == "1"
        self.capture_error = None
        self.graph = None
        self.static_batch = None
        self.static_out = None
        self._capture_stream = None
        self.seen = 0

    def run(self, batch: Dict[str, torch.Tensor], step_fn: Callable, optimizers: Iterable):
        """``step_fn(batch) -> dict of device tensors`` performs the whole iteration; returns that dict (static
        tensors once the graph is live: read them before the next call)."""
        if not self.enabled:
            return step_fn(batch)
        if self.graph is None:
            if self.seen < self.warmup:
                self.seen += 1
                return step_fn(batch)
            try:
                self._capture(batch, step_fn, optimizers)
            except Exception as err:  # a library call that cannot be captured: stay eager
                # No captured kernel has executed.  What _capture() did before the failure is harmless for an eager
                # step: push_hyper() only refreshed the device copy of the hyper-parameters, and the gradients it set to
                # None are re-created by the step's own zero_grad() / backward.
                self.capture_error = repr(err)
                if self.strict:
                    raise
                import warnings
                warnings.warn(f"hipGraph capture of the training step failed ({err!r}); continuing eagerly")
                self.enabled, self.graph = False, None
                torch.cuda.synchronize()
                return step_fn(batch)
        if not self._matches(batch):
            # e.g. the last, smaller batch of an epoch: a copy into the static inputs would raise (or silently broadcast a
            # batch of one over the captured batch size), so this iteration runs eagerly; the graph stays valid (its
            # captured launches hold their own gradient / table addresses, re-uploaded on every replay)
            return step_fn(batch)
        for key, value in batch.items():
            self.static_batch[key].copy_(value, non_blocking=True)
        for opt in optimizers:
            opt.push_hyper()
        self.graph.replay()
        return self.static_out

    def _matches(self, batch) -> bool:
        if batch.keys() != self.static_batch.keys():
            return False
        return all(value.shape == self.static_batch[key].shape and value.dtype == self.static_batch[key].dtype
                   and value.device == self.static_batch[key].device for key, value in batch.items())

    def _capture(self, batch, step_fn, optimizers):
        self.static_batch = {key: value.clone() for key, value in batch.items()}
        for opt in optimizers:
            if not hasattr(opt, "push_hyper"):
                raise RuntimeError("StepGraph needs optimizers with device-side hyper-parameters (FusedSGD)")
            opt.push_hyper()
            opt.zero_grad(set_to_none=True)  # gradients are re-created inside the graph's memory pool
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # With an RCCL process group in the process its watchdog thread polls the events of earlier (eager) collectives; under
        # the default "global" capture mode an event query from ANY thread while a capture is open is an error
        # (hipErrorStreamCaptureUnsupported) and the watchdog takes the process down.  "thread_local" restricts the check to
        # the capturing thread, which is what the capture needs; work objects created DURING the capture are never handed to
        # the watchdog (training/grad_exchange.py issues them from the capturing stream).
        mode = "global"
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
                mode = "thread_local"
        except Exception:
            pass
        # Per-stream kernel state (completion counters, split-K scratch) is created and zeroed on the capture stream as EAGER
        # work before the capture opens; buffers first created inside a capture that then fails are forgotten again.
        import sis_hip
        device = next(iter(self.static_batch.values())).device
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(device)
        self._capture_stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self._capture_stream):
            snapshot = sis_hip.prepare_capture(device)
        self._capture_stream.synchronize()
        try:
            with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode=mode):
                self.static_out = step_fn(self.static_batch)
        except Exception:
            sis_hip.rollback_capture(snapshot)
            raise
        self.graph = graph
