"""Data-parallel gradient exchange for the segmentation training step: bucketed reduce-scatter + all-gather (or one
all-reduce) on RCCL's stream, overlapped with backward, capturable inside the step's hipGraph.

Role in the reference: ``DistributedDataParallel(network, device_ids=[rank], find_unused_parameters=...,
broadcast_buffers=False)`` (training_builder/base_train_builder.py:40-43), whose C++ reducer all-reduces gradient buckets
while backward runs.  torch's reducer stays available (``data_parallel: ddp`` in the config); this module is the default
because three things of the MI355X step do not fit it:

* **the whole iteration is one hipGraph** (training/graph_step.py).  DDP's reducer rebuilds its buckets on the second
  iteration, copies a used-parameter bitmap to the host when ``find_unused_parameters`` is set (EMANet needs it:
  ``emau.conv1`` never receives a gradient) and wants eleven eager iterations before a capture.  Here the bucket plan is
  fixed after ONE discovery backward (the parameters that received a gradient, in the order they became ready: unused
  parameters simply never appear, their ``.grad`` stays ``None`` and the optimizer skips them as ``torch.optim.SGD`` does),
  every later backward issues the same launches in the same order, and the collectives are ordinary stream work that the
  capture records like any kernel.
* **gradients must sit at fixed addresses**: ``FusedSGD`` (training/fused_sgd.py) reads them through a device pointer
  table.  Buckets are persistent flat fp32 buffers, registered with ``sis_hip`` as the gradient arena: the weight-gradient
  kernels (convolutions, Linear layers, the weight-standardisation bank: nearly all bytes) write their result INTO the
  parameter's slice and hand autograd a view of it, so those gradients are born in place; when the last gradient of a
  bucket has arrived ONE fused copy (``torch._foreach_copy_``) moves the small rest (norm parameters, biases) and
  ``.grad`` is re-pointed at the bucket views, so the table is uploaded once.
* **xGMI is point-to-point** (7 links per GPU): by default a bucket is reduced as ``reduce_scatter`` (every
  rank owns 1/N of the bucket) followed by ``all_gather``, each on the bucket's own slice of the flat buffer (in place),
  issued straight into librccl.so (``ncclReduceScatter`` / ``ncclAllGather`` on the process group's communicator) on the
  exchange's own HIP stream at world size 1 (the only configuration a one-GPU box can run and verify) and, until a
  multi-rank run has exercised it, through torch.distributed at world size > 1 unless SIS_DP_DIRECT_RCCL=1 (then behind a
  start-up self-check against torch's all_reduce);
  ``collective: allreduce`` issues one ``all_reduce`` instead.  Averaging is the collective's own ``AVG`` on RCCL; gloo
  (CPU rehearsals) sums and scales.

Semantics kept from the reference wrap: parameters are broadcast from rank 0 at construction, buffers are NOT synchronised
(``broadcast_buffers=False``: batch-norm statistics and EMANet's ``emau.mu`` stay per rank), gradients are the mean over
ranks, ``.module`` is the wrapped network.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


_DEBUG = os.environ.get("SIS_DP_DEBUG", "0") == "1"
# Collectives straight into librccl.so (ctypes on the communicator ProcessGroupNCCL owns: a few microseconds of host time each,
# stream work only) or through torch.distributed's work objects (SIS_DP_DIRECT_RCCL=0).  Measured at world size 1, TransUNet,
# 19 buckets = 38 collectives per iteration: 25.4 ms per eager iteration direct against 25.9 ms through torch.distributed on
# one box, 25.0 against 34.1 ms on another (the host cost of 38 c10d calls lands on an iteration whose 860 launches already keep
# the host busy: a slower host falls behind); EMANet, 7 buckets: 28.3 / 28.3 and 28.4 / 28.5 ms.
# Capturing the collectives into the step hipGraph is a separate decision (``capturable``): by default only at world size 1 --
# the one configuration a 1-GPU box can run (RCCL refuses two ranks on one device) and tests/test_distributed_gpu.py +
# bench.py's data_parallel_rehearsal verify; SIS_DP_GRAPH=1 extends it to any world size.  Nothing is lost by staying eager
# there: the wrapped iteration is as fast eager as captured (25.0 / 25.2 ms, 28.4 / 28.3 ms).
# SIS_DP_DIRECT_RCCL: "auto" (default) = direct calls at world size 1 -- the configuration a one-GPU box runs, tests and
# captures -- and torch.distributed's collectives at world size > 1, where the direct path (shard offsets, in-place
# reduce-scatter / all-gather aliasing, RCCL enum values) has never executed on hardware; "1" = direct at any
# world size, after a start-up self-check of one bucket-shaped exchange against torch's all_reduce (a failed check falls back
# and says so); "0" = never.
_DIRECT_RCCL = os.environ.get("SIS_DP_DIRECT_RCCL", "auto")
_DP_GRAPH = os.environ.get("SIS_DP_GRAPH", "auto")


def shard_plan(numel: int, world: int):
    """Padded length and per-rank shard of a bucket holding ``numel`` gradient elements: the flat buffer is padded to a multiple
    of 4 * world so that every rank's shard has the same length AND starts 16-byte aligned -> (padded_numel, per_rank)."""
    if numel <= 0 or world <= 0:
        raise ValueError("shard_plan needs a positive element count and world size")
    quantum = 4 * world
    padded = (numel + quantum - 1) // quantum * quantum
    return padded, padded // world


def shard_span(rank: int, per: int):
    """Element range [begin, end) of ``rank``'s shard inside a bucket: the in-place convention of ncclReduceScatter
    (recvbuff = sendbuff + rank * recvcount) and ncclAllGather (sendbuff = recvbuff + rank * sendcount)."""
    return rank * per, (rank + 1) * per


class _Rccl:
    """The RCCL entry points the exchange needs, called on the process group's own communicator (``ProcessGroupNCCL._comm_ptr``)
    with plain pointers and a HIP stream -- the C ABI of librccl.so, the library torch itself links: no torch work objects, no
    watchdog bookkeeping, nothing but stream work, which is what a hipGraph capture wants.  (Through torch.distributed the
    collectives captured fine, but the process group's watchdog thread intermittently queried an event that had last been
    recorded while capturing -- hipErrorCapturedEvent -- and took the process down.)"""
    FLOAT32, SUM, AVG = 7, 0, 4   # ncclDataType_t / ncclRedOp_t values (nccl.h)
    _lib = None
    path = None

    @classmethod
    def lib(cls):
        """The librccl.so instance ALREADY MAPPED into this process (the one torch's ProcessGroupNCCL created the communicator
        with), found through /proc/self/maps and opened with RTLD_NOLOAD: handing torch's ncclComm_t to a second copy of the
        library (a torch built against /opt/rocm's RCCL, say) would be undefined behaviour, and a missing file must not
        surface as an OSError inside the first backward.  None when no mapped RCCL is found."""
        if cls._lib is None:
            import ctypes
            path = None
            try:
                with open("/proc/self/maps") as maps:
                    for line in maps:
                        name = line.rsplit(" ", 1)[-1].strip()
                        if "/" in name and os.path.basename(name).startswith(("librccl.so", "libnccl.so")):
                            path = name
                            break
            except OSError:
                path = None
            if path is None:
                cls._lib = False
                return None
            try:
                lib = ctypes.CDLL(path, mode=getattr(os, "RTLD_NOLOAD", 4) | getattr(os, "RTLD_NOW", 2))
                vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
                lib.ncclReduceScatter.argtypes, lib.ncclReduceScatter.restype = [vp, vp, sz, i, i, vp, vp], i
                lib.ncclAllGather.argtypes, lib.ncclAllGather.restype = [vp, vp, sz, i, vp, vp], i
                lib.ncclAllReduce.argtypes, lib.ncclAllReduce.restype = [vp, vp, sz, i, i, vp, vp], i
                lib.ncclGetErrorString.argtypes, lib.ncclGetErrorString.restype = [i], ctypes.c_char_p
                cls._lib, cls.path = lib, path
            except (OSError, AttributeError):
                cls._lib = False
        return cls._lib or None

    @classmethod
    def check(cls, rc, what):
        if rc != 0:
            raise RuntimeError(f"RCCL {what} failed: {cls.lib().ncclGetErrorString(rc).decode()}")

    @staticmethod
    def communicator(process_group, device):
        """ncclComm_t of the group's RCCL backend on ``device`` as an integer, or None when torch does not expose it."""
        try:
            group = process_group if process_group is not None else dist.distributed_c10d._get_default_group()
            backend = group._get_backend(device)
            ptr = int(backend._comm_ptr())
            return ptr or None
        except Exception:
            return None


class _Bucket:
    __slots__ = ("params", "views", "offsets", "flat", "numel", "per", "pending", "index")

    def __init__(self, index: int, params: List[torch.Tensor], world: int, device, dtype):
        self.index, self.params = index, params
        # every parameter's slice starts 16-byte aligned (4 fp32 elements): the weight-gradient kernels that write into the
        # slices store 16 bytes per lane; the gaps stay zero on every rank and ride through the collectives as zeros
        self.offsets, at = [], 0
        for p in params:
            self.offsets.append(at)
            at += (p.numel() + 3) // 4 * 4
        self.numel, self.per = shard_plan(at, world)   # every rank's shard starts 16-byte aligned as well
        self.flat = torch.zeros(self.numel, dtype=dtype, device=device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, params)]
        self.pending = len(params)


class BucketedDataParallel(nn.Module):
    """``BucketedDataParallel(network)``: same call surface as the wrapped network; gradients of every backward are averaged
    over the process group's ranks before ``backward()`` returns (stream-ordered on a HIP device)."""

    def __init__(self, module: nn.Module, process_group=None, bucket_cap_mb: float = 25.0, collective: Optional[str] = None):
        super().__init__()
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("BucketedDataParallel needs an initialised torch.distributed process group")
        self.module = module
        self.process_group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.backend = dist.get_backend(process_group)
        self.collective = collective or os.environ.get("SIS_GRAD_COLLECTIVE", "rs_ag")
        if self.collective not in ("rs_ag", "allreduce"):
            raise ValueError("collective must be 'rs_ag' (reduce-scatter + all-gather) or 'allreduce'")
        self.bucket_bytes = int(bucket_cap_mb * (1 << 20))
        self._params = [p for p in module.parameters() if p.requires_grad]
        if not self._params:
            raise RuntimeError("BucketedDataParallel: the network has no parameter that requires a gradient")
        self.device = self._params[0].device
        self._on_gpu = self.device.type == "cuda"
        if self._on_gpu and self.backend == "gloo":
            # gloo stages device tensors through the host and synchronises: correct (one-GPU rehearsals), never capturable
            pass
        self._pending = []   # work handles of collectives issued through torch.distributed during the current backward
        self._postponed = []   # (world size 1) buckets complete but for gradients still in sis_hip's deferred queue
        if self._on_gpu and self.world > 1:
            import sis_hip
            sis_hip.block_wgrad_deferral(self)   # buckets leave during the backward: no gradient waits for its end
        self._comm, self._comm_stream, self._joined = None, None, True
        self.direct_rccl_note = None   # why the direct path is off when it was wanted (bench.py prints it)
        want_direct = _DIRECT_RCCL == "1" or (_DIRECT_RCCL == "auto" and self.world == 1)
        if self._on_gpu and self.backend == "nccl" and want_direct:
            if _Rccl.lib() is None:
                self.direct_rccl_note = "no librccl.so mapped into this process"
            else:
                self._comm = _Rccl.communicator(process_group, self.device)
                if self._comm is None:   # the communicator is created lazily by the first collective
                    probe = torch.zeros(1, device=self.device)
                    dist.all_reduce(probe, group=process_group)
                    torch.cuda.synchronize(self.device)
                    self._comm = _Rccl.communicator(process_group, self.device)
                if self._comm is None:
                    self.direct_rccl_note = "this torch build does not expose ProcessGroupNCCL._comm_ptr"
                else:
                    import sis_hip
                    self._comm_stream = sis_hip.side_stream(self.device)   # (on a hardware queue of its own: the exchange overlaps the backward)
                    if self.world > 1 and not self._self_check():
                        self.direct_rccl_note = "start-up self-check of the direct reduce-scatter + all-gather failed"
                        self._comm, self._comm_stream = None, None
            if self.direct_rccl_note and self.rank == 0:
                import warnings
                warnings.warn(f"BucketedDataParallel: collectives go through torch.distributed ({self.direct_rccl_note})")
        self.buckets: Optional[List[_Bucket]] = None
        self._bucket_of = {}
        self._order: List[torch.Tensor] = []      # discovery: parameters in the order their gradients became ready
        self._callback_queued = False
        self._flushed = 0
        # copied_elems / in_place_elems: gradient elements the gather had to copy / found already written into their bucket slice
        self.stats = {"backwards": 0, "collectives": 0, "discovery_backwards": 0, "copied_elems": 0, "in_place_elems": 0}
        self._broadcast_parameters()
        for p in self._params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- construction ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _broadcast_parameters(self):
        """Rank 0's parameters everywhere (what DDP's constructor does); buffers stay as each rank made them."""
        if self.world == 1:
            return
        chunk, size = [], 0
        def send(tensors):
            flat = torch.cat([t.detach().reshape(-1) for t in tensors])
            dist.broadcast(flat, src=dist.get_global_rank(self.process_group, 0) if self.process_group is not None else 0,
                           group=self.process_group)
            at = 0
            for t in tensors:
                t.detach().copy_(flat[at:at + t.numel()].view_as(t))
                at += t.numel()
        for p in self.module.parameters():
            if chunk and (p.dtype != chunk[0].dtype or size + p.numel() * p.element_size() > (64 << 20)):
                send(chunk)
                chunk, size = [], 0
            chunk.append(p)
            size += p.numel() * p.element_size()
        if chunk:
            send(chunk)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _rccl_exchange(self, flat: torch.Tensor, numel: int, per: int, stream):
        """The direct form of one bucket's exchange on ``stream`` (a torch.cuda.Stream): in-place reduce-scatter into this rank's
        shard followed by the all-gather of the shards, or one all-reduce; ``AVG`` on RCCL.  At world size 1 the mean of one
        rank is that rank, and the reduction is issued as ``SUM``: RCCL's one-rank AVG is a pre-multiply kernel that reads and
        rewrites the whole bucket (``oneRankReduce<FuncPreMulSum>``: 1.15 ms per TransUNet iteration at 0.74 TB/s,
        profiles/r05_transunet_dp_step_breakdown.txt), its one-rank in-place SUM is nothing at all."""
        import ctypes
        R = _Rccl
        lib = R.lib()
        op = R.SUM if self.world == 1 else R.AVG
        st = ctypes.c_void_p(stream.cuda_stream)
        comm, base = ctypes.c_void_p(self._comm), flat.data_ptr()
        with torch.cuda.device(self.device):
            if self.collective == "rs_ag":
                begin, _ = shard_span(self.rank, per)
                mine = ctypes.c_void_p(base + 4 * begin)
                R.check(lib.ncclReduceScatter(ctypes.c_void_p(base), mine, per, R.FLOAT32, op, comm, st), "ncclReduceScatter")
                R.check(lib.ncclAllGather(mine, ctypes.c_void_p(base), per, R.FLOAT32, comm, st), "ncclAllGather")
            else:
                R.check(lib.ncclAllReduce(ctypes.c_void_p(base), ctypes.c_void_p(base), numel, R.FLOAT32, op, comm, st), "ncclAllReduce")

    @torch.no_grad()
    def _self_check(self) -> bool:
        """World size > 1 with the direct path requested: one bucket-shaped exchange of rank-dependent values through the direct
        calls must equal torch.distributed's all_reduce(AVG) of the same values on every rank -- shard offsets, in-place
        aliasing and the enum values are then verified on THIS machine before any gradient depends on them."""
        numel, per = shard_plan(4099, self.world)
        gen = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        ref = torch.randn(numel, generator=gen).to(self.device)
        got = ref.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.AVG, group=self.process_group)
        self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
        self._rccl_exchange(got, numel, per, self._comm_stream)
        torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)
        ok = torch.tensor([1.0 if torch.allclose(got, ref, rtol=1e-5, atol=1e-6) else 0.0], device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.process_group)   # every rank takes the same branch
        return bool(ok.item() == 1.0)

    # ---- backward ----------------------------------------------------------------------------------------------------
    def _on_grad(self, param):
        if not self._callback_queued:
            # runs once the engine has finished THIS backward (the mechanism DDP's reducer and FSDP use)
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
            self._callback_queued = True
        if self.buckets is None:
            self._order.append(param)
            return
        bucket = self._bucket_of.get(id(param))
        if bucket is None:
            raise RuntimeError("BucketedDataParallel: a parameter that had no gradient in the first backward received one now; "
                               "the bucket plan is fixed after the first iteration (rebuild the wrapper)")
        bucket.pending -= 1
        if bucket.pending == 0:
            self._flush(bucket)

    @torch.no_grad()
    def _gather(self, bucket: _Bucket):
        """Gradients of the bucket -> its flat buffer, ``.grad`` -> views.  Gradients the weight-gradient kernels wrote straight into
        their slice (``sis_hip.grad_out``: the convolution / Linear weights, i.e. nearly all bytes) are already in place; one
        fused copy moves the rest (norm parameters, biases, anything a library kernel produced)."""
        src, dst = [], []
        for p, view in zip(bucket.params, bucket.views):
            g = p.grad
            if g is None:
                raise RuntimeError("BucketedDataParallel: a bucket was flushed before all of its gradients existed")
            if g.data_ptr() != view.data_ptr():
                src.append(g if g.dtype == view.dtype and g.is_contiguous() else g.to(view.dtype).contiguous())
                dst.append(view)
                self.stats["copied_elems"] += view.numel()
            else:
                self.stats["in_place_elems"] += view.numel()
        if src:
            torch._foreach_copy_(dst, src)
            for p, view in zip(bucket.params, bucket.views):
                p.grad = view

    @torch.no_grad()
    def _reduce(self, bucket: _Bucket):
        """Issues the bucket's collective(s) asynchronously FROM the stream backward is running on: the process group's own
        RCCL stream first waits for that stream (the gathered bucket is complete there), then runs the collective beside the
        rest of backward; ``_finish`` makes the compute stream wait for the work handles.  Called from the capturing stream
        during a hipGraph capture, so that the process group sees the capture and keeps these work objects away from its
        watchdog thread (an event recorded in a capturing stream must not be queried; launching from a side stream of our own
        let the watchdog do exactly that)."""
        flat, world = bucket.flat, self.world
        self.stats["collectives"] += 1
        if _DEBUG and self._on_gpu:
            import threading
            print(f"[grad_exchange] bucket {bucket.index}: thread {threading.current_thread().name}, stream "
                  f"{torch.cuda.current_stream(self.device).cuda_stream:#x}, capturing {torch.cuda.is_current_stream_capturing()}", flush=True)
        if self.backend == "nccl" and self._comm is not None:
            # straight into RCCL on the exchange's own stream: the gathered bucket is complete on the compute stream (event),
            # backward keeps running there meanwhile; ``_finish`` joins the streams
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            self._rccl_exchange(flat, bucket.numel, bucket.per, self._comm_stream)
            self._joined = False
        elif self.backend == "nccl":
            avg = dist.ReduceOp.AVG
            if self.collective == "rs_ag":
                begin, end = shard_span(self.rank, bucket.per)
                shard = flat[begin:end]
                self._pending.append(dist.reduce_scatter_tensor(shard, flat, op=avg, group=self.process_group, async_op=True))
                self._pending.append(dist.all_gather_into_tensor(flat, shard, group=self.process_group, async_op=True))
            else:
                self._pending.append(dist.all_reduce(flat, op=avg, group=self.process_group, async_op=True))
        else:   # gloo: no AVG, no reduce_scatter_tensor
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
            flat.mul_(1.0 / world)

    def _flush(self, bucket: _Bucket, at_end=False):
        if self._on_gpu:
            import sis_hip
            if self.world == 1 and not at_end and sis_hip.deferred_wgrad_pending():
                # One rank has nobody to overlap with: the bucket waits for the end of the backward, where the queued weight
                # gradients of ALL layers run as one launch per Linear shape (a flush here would multiply this bucket's layers alone).
                self._postponed.append(bucket)
                return
            sis_hip.flush_deferred()   # gradients whose second-stage reduction was deferred are completed before they travel
        self._gather(bucket)
        self._reduce(bucket)
        self._flushed += 1

    @torch.no_grad()
    def _finish(self):
        """End of a backward: discovery -> fix the plan and reduce everything now; otherwise join the side stream."""
        self._callback_queued = False
        self.stats["backwards"] += 1
        if self.buckets is None:
            self.stats["discovery_backwards"] += 1
            self._plan()
            for bucket in self.buckets:
                self._flush(bucket)
        postponed, self._postponed = self._postponed, []
        for bucket in postponed:
            self._flush(bucket, at_end=True)
        if self._flushed != len(self.buckets):
            missing = [b.index for b in self.buckets if b.pending != 0]
            for work in self._pending:       # leave no collective in flight behind the exception
                work.wait()
            self._pending = []
            if not self._joined:
                torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)
                self._joined = True
            self._reset()
            raise RuntimeError(f"BucketedDataParallel: buckets {missing} did not receive all of their gradients in this backward "
                               "(a parameter used in the first iteration was unused now)")
        for work in self._pending:
            work.wait()        # stream-ordered on a HIP device: the compute stream waits for RCCL's; optimizer.step() reads the buckets
        self._pending = []
        if not self._joined:
            torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)   # optimizer.step() reads the buckets
            self._joined = True
        self._reset()

    def _reset(self):
        self._flushed = 0
        if self._on_gpu:
            try:
                import sis_hip
                sis_hip.grad_arena_reset(self)   # every bucket slice may be handed to a weight-gradient kernel again
            except ImportError:
                pass
        if self.buckets is not None:
            for b in self.buckets:
                b.pending = len(b.params)

    def _plan(self):
        """Buckets of <= bucket_cap_mb in readiness order (the first bucket is the first to be complete in every later
        backward).  Every rank runs the same graph, hence builds the same plan; checked by comparing a digest."""
        seen, order = set(), []
        for p in self._order:
            if id(p) not in seen:
                seen.add(id(p))
                order.append(p)
        self._order = []
        if not order:
            raise RuntimeError("BucketedDataParallel: backward produced no parameter gradient")
        # Parameters whose gradients one kernel writes as ONE stacked tensor (modules announce them through
        # ``grad_fusion_groups()``: the ViT encoder's query | key | value weights) are kept together, in the announced order, at
        # the position of the first of them to become ready, and never split over two buckets.
        fused = {}
        for m in self.module.modules():
            if hasattr(m, "grad_fusion_groups"):
                for group in m.grad_fusion_groups():
                    if all(id(p) in seen for p in group):
                        for p in group:
                            fused[id(p)] = group
        units, placed = [], set()
        for p in order:
            if id(p) in placed:
                continue
            unit = list(fused.get(id(p), (p,)))
            placed.update(id(q) for q in unit)
            units.append(unit)
        groups, cur, size = [], [], 0
        for unit in units:
            nbytes = sum(p.numel() for p in unit) * 4
            if cur and size + nbytes > self.bucket_bytes:
                groups.append(cur)
                cur, size = [], 0
            cur += unit
            size += nbytes
        groups.append(cur)
        self.buckets = [_Bucket(i, g, self.world, self.device, torch.float32) for i, g in enumerate(groups)]
        self._bucket_of = {id(p): b for b in self.buckets for p in b.params}
        if self._on_gpu:
            try:   # weight-gradient kernels write into the slices from now on (no gather copy for those parameters)
                import sis_hip
                for b in self.buckets:
                    sis_hip.grad_arena_register(self, b.params, [b.flat] * len(b.params), b.offsets)
            except ImportError:
                pass
        if self.world > 1:
            # what must agree across ranks: bucket count and sizes, and WHICH parameter sits where (a rolling hash of the
            # parameters' positions in module order and their element counts, bucket by bucket)
            position = {id(p): i for i, p in enumerate(self.module.parameters())}
            def folded(values):
                h = 1469598103934665603
                for v in values:
                    h = ((h ^ int(v)) * 1099511628211) & 0x7FFFFFFFFFFFFFFF
                return h
            # a fixed-length record (ranks with different bucket counts must still be able to compare): counts, total padded
            # length, a hash of the per-bucket lengths and a hash of (parameter position, element count) in bucket order
            words = [len(self.buckets), len(order), sum(b.numel for b in self.buckets), folded(b.numel for b in self.buckets),
                     folded(v for b in self.buckets for p in b.params for v in (position[id(p)], p.numel()))]
            digest = torch.tensor(words, dtype=torch.int64, device=self.device if self.backend == "nccl" else "cpu")
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.process_group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.process_group)
            if not torch.equal(lo, hi):
                raise RuntimeError("BucketedDataParallel: ranks disagree on the bucket plan (different graphs per rank)")

    def __del__(self):
        try:
            import sis_hip
            sis_hip.grad_arena_release(self)
            sis_hip.block_wgrad_deferral(self, False)
        except Exception:
            pass

    # ---- introspection (tests, bench) --------------------------------------------------------------------------------
    def direct_rccl(self) -> bool:
        """The collectives go straight into librccl.so on the group's communicator (no torch.distributed work objects)."""
        return self._comm is not None

    def bucket_spans(self):
        return [(b.flat.data_ptr(), b.flat.data_ptr() + 4 * b.numel) for b in (self.buckets or [])]

    def capturable(self) -> bool:
        """The step hipGraph may include this exchange: collectives issued straight into RCCL are stream work; torch.distributed's
        work objects (watchdog events) and gloo (host synchronisation) are not.  At world size > 1 only with SIS_DP_GRAPH=1."""
        return (self._on_gpu and self.backend == "nccl" and self._comm is not None
                and (_DP_GRAPH == "1" or (_DP_GRAPH == "auto" and self.world == 1)))
