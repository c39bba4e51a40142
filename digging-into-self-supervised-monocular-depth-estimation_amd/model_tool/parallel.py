"""`grad_sync`: the one exchange of a data-parallel step -- all-reduce(mean) of the gradients over the GPUs of a node
(RCCL over xGMI: torch.distributed backend "nccl" on ROCm) -- built so that the WHOLE step, exchange included, can be
captured into one hipGraph (model_train.graphed_step).  The reference is single-device (model_train.py:28,54-96).

Why not torch's DistributedDataParallel (round 2 wrapped each of the four networks in one): its reducer keeps
host-side state per backward (bucket bookkeeping, unused-parameter logic, four sets of autograd hooks here) -- the
captured step had to be switched off whenever WORLD_SIZE > 1, and an eager step of this model is host-bound with the
loader's workers alive (~1600 launches).  Here:

  * ONE flat float32 buffer holds every trainable parameter's gradient, allocated once (static addresses), cut into
    buckets along the order in which backward produces gradients (reverse registration order: pose networks, depth
    decoder, depth encoder);
  * autograd writes its gradients where it likes (`.grad` is None before backward, so the accumulation node keeps the
    tensor the last backward kernel produced: no extra pass).  A post-accumulate hook counts a bucket's parameters
    down; when the last one is written the bucket's gradients are gathered into the flat buffer by ONE multi-tensor
    copy, `.grad` of its parameters is re-pointed at the views (what Adam will read) and the bucket's all-reduce is
    issued -- asynchronously, on the communicator's stream;
  * `finish()` issues what no hook issued (a parameter that received no gradient contributes zeros), makes the current
    stream wait for every exchange and, where the backend has no AVG reduction (gloo), divides by the world size.
    No host synchronisation, no per-step Python state that a graph replay would skip.

Measured on one MI355X with a process group of one rank (profiles/r03_dp_step.txt): gathering into the flat buffer
costs what the in-place accumulation of round 3's first version did not (that version pre-set `.grad` to the views:
+0.39 ms = 250 read-modify-write kernels and a 107 MB memset per step).  EAGER, four 32 MB buckets overlap backward at
no cost.  CAPTURED, every bucket that overlaps backward is a fork / join in the hipGraph and costs ~0.37 ms of graph
execution (four buckets: +1.1 ms against one) -- more than the ~0.25 ms of xGMI time it could hide -- so a captured
step uses ONE bucket, issued when backward ends (`setting` picks bucket_mb accordingly).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the 107 MB (ResNet-18 depth + pose, fp32) is
link-bound at ~0.5-1 ms against a >=13 ms step.  `comm_dtype=torch.bfloat16` halves the bytes (gradients are rounded
once before the sum; off by default).

Collectives and stream capture (the round-3 core dump): a collective issued SYNCHRONOUSLY (async_op=False)
inside `torch.cuda.graph` registers its work with ProcessGroupNCCL's watchdog thread, which then polls an event that was
recorded in the capturing stream -- hipErrorCapturedEvent, std::terminate, the whole rank gone.  Only an `async_op=True`
collective whose wait() is a stream-side wait (what grad_sync issues) may be captured.  Every other collective of this
module refuses to run while the current stream is capturing (`_refuse_under_capture`): a RuntimeError in the caller
instead of an abort in another thread.

Contract of grad_sync: exactly ONE backward between zero() and finish().  A second one (gradient accumulation,
retain_graph) would add local gradients into the already exchanged views; the hook raises instead."""
import os

import torch
import torch.distributed as dist


def dp_graph_allowed(world):
    """May the step of a `world`-rank job be captured into ONE hipGraph, exchange included?  (If not, a captured step takes the
    SPLIT form -- forward + backward + gather in one graph, the all-reduce issued eagerly, Adam in a second graph -- which keeps
    RCCL out of the capture: model_train.graphed_step.)  One rank: yes (measured and
    tested on the GPU: tests/test_gpu_driver.py).  Several ranks: only with MDX_DP_GRAPH=1 -- a captured MULTI-rank RCCL
    all-reduce (fork / join of the communicator's stream inside the capture, replay order across ranks) has not run on
    hardware yet (no multi-GPU node was available to this build: SCALE_r01..r03 were skipped), so the default
    data-parallel step is the eager one with 32 MB buckets overlapping backward.  tools/scale_check.sh holds the runs
    that settle it."""
    return int(world) <= 1 or os.environ.get("MDX_DP_GRAPH", "") == "1"


def capturing():
    """True while the current HIP stream is being captured into a graph."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _refuse_under_capture(what):
    if capturing():
        raise RuntimeError(
            "%s is a synchronous collective and the current stream is being captured into a hipGraph: the process "
            "group's watchdog would query an event recorded inside the capture (hipErrorCapturedEvent) and abort the "
            "rank.  Call it outside torch.cuda.graph(); only grad_sync's asynchronous all-reduce is capturable." % what)


def grad_view(flat, offset, p):
    """The slice of the flat buffer that serves as `p.grad`: same shape AND strides as the parameter (a channels_last
    weight gets a channels_last gradient -- torch's fused Adam requires params, grads and moments to share one layout and
    pairs elements by storage order otherwise)."""
    expect = 1
    for st, sz in sorted((st, sz) for st, sz in zip(p.stride(), p.shape) if sz != 1):
        if st != expect:                         # non-overlapping and dense: a permutation of the contiguous strides
            raise ValueError("grad_sync: parameter of shape %s has strides %s (not dense)" % (tuple(p.shape), p.stride()))
        expect *= sz
    # the parameter's strides verbatim, size-1 dimensions included: the fused optimiser compares stride tuples
    return flat.as_strided(p.shape, p.stride(), storage_offset=flat.storage_offset() + offset)


def _align(n, a=64):
    return (n + a - 1) // a * a


class grad_sync(object):
    no_comm = False       # measurement aid (profiles/r03_dp_step.txt "nocomm" rows): everything but the collective itself

    def __init__(self, parameters, bucket_mb=32, group=None, comm_dtype=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("grad_sync: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("grad_sync: parameters must be float32 on one device")
        order = list(reversed(self.params))                      # ~ the order backward writes them
        cap = max(1, int(bucket_mb * (1 << 20) // 4))
        self.offsets, self.buckets, self._bucket_of = {}, [], {}
        off = start = 0
        members = []
        for p in order:
            self.offsets[id(p)] = off
            members.append(p)
            off += _align(p.numel())                             # 256-byte aligned views
            if off - start >= cap:
                self.buckets.append((start, off, members))
                start, members = off, []
        if members:
            self.buckets.append((start, off, members))
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.comm = torch.zeros(off, dtype=comm_dtype, device=dev) if comm_dtype not in (None, torch.float32) else None
        self._views = {}
        for k, (_, _, members) in enumerate(self.buckets):
            for p in members:
                self._bucket_of[id(p)] = k
                self._views[id(p)] = grad_view(self.flat, self.offsets[id(p)], p)
        self._avg = self.backend == "nccl"                      # ncclAvg exists in RCCL; gloo only sums
        self._works = []
        self._xs, self._streams = None, []                       # exchange stream; streams seen producing gradients
        self._reset()
        self.zero()
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in self.params]

    def backward_streams(self, streams):
        """Side streams on which parts of backward run (besides the stream backward() is called on): the exchange waits for them
        too.  Static, so that the per-parameter hook stays as light as it can be (an eager data-parallel step is host-bound)."""
        self._streams = [t for t in streams if t is not None]

    def zero(self):
        """Instead of optimizer.zero_grad(): `.grad` = None, so backward's accumulation keeps the incoming tensor as
        it is (no read-modify-write pass); the gather into the flat buffer overwrites, nothing needs clearing."""
        for p in self.params:
            p.grad = None

    # -- exchange ---------------------------------------------------------------------------------
    def _reset(self):
        self._pending = [len(m) for _, _, m in self.buckets]
        self._next = 0                     # buckets are issued strictly in order: every rank issues the same sequence
        self._keep = []                    # gathered gradients stay alive until finish(): their memory must not be handed out
                                           # again while the exchange stream still reads them

    def _ready(self, p):
        k = self._bucket_of[id(p)]
        if k < self._next:
            # the bucket is on the wire (or back) and `.grad` is the exchanged view: this gradient comes from a SECOND
            # backward since zero() and was accumulated, rank-locally, into the reduced values
            raise RuntimeError("grad_sync: a gradient arrived after its bucket was exchanged -- more than one backward "
                               "between zero() and finish() (gradient accumulation / retain_graph are not supported)")
        self._pending[k] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._issue(self._next)
            self._next += 1

    def _issue(self, k):
        if not self.flat.is_cuda:
            return self._issue_on_current(k)
        # On the exchange stream, behind the backward streams' work so far: the streams that run backward are never made to wait for
        # each other or for the gather (with whole-stream waits on the issuing stream the two backward streams serialised: 634
        # against 654 images/s without the overlap; stream-side waits on events are capturable).
        if self._xs is None:
            self._xs = torch.cuda.Stream(self.flat.device)
        # everything the backward streams have been given so far, the bucket's gradients included: the stream this hook runs on and
        # the ones the trainer registered (the pose network's backward runs beside the depth network's, model_train.trainer)
        self._xs.wait_stream(torch.cuda.current_stream(self.flat.device))
        for t in self._streams:
            self._xs.wait_stream(t)
        with torch.cuda.stream(self._xs):
            self._issue_on_current(k)

    def _issue_on_current(self, k):
        a, b, members = self.buckets[k]
        dst, src = [], []
        for p in members:
            v = self._views[id(p)]
            if p.grad is None:
                v.zero_()                  # no gradient this step (never on this model's default path)
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)                       # one multi-tensor kernel
            self._keep.extend(src)
        for p in members:
            p.grad = self._views[id(p)]                          # what the optimiser reads: the reduced values
        buf = self.flat[a:b]
        if self.comm is not None:
            cbuf = self.comm[a:b]
            cbuf.copy_(buf)
            buf = cbuf
        if self.no_comm:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        # async_op=True is what makes this capturable (module docstring): never change it to a blocking call
        work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        assert work is not None, "grad_sync: the all-reduce must be asynchronous"
        self._works.append((k, work))

    def gather_only(self):
        """Context manager around backward() + finish() of a step whose exchange happens OUTSIDE a captured graph
        (model_train.graphed_step, split form): buckets are gathered into the flat buffer as they complete, nothing is sent --
        also not by the hook of a bucket's last gradient, which fires INSIDE backward.  Pair with exchange()."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            was, self.no_comm = self.no_comm, True
            try:
                yield self
            finally:
                self.no_comm = was
        return scope()

    def exchange(self):
        """The whole flat buffer in ONE all-reduce(mean), issued eagerly on the current stream (no host synchronisation on
        RCCL): what sits between the two graphs of the split form."""
        buf = self.flat
        if self.comm is not None:
            self.comm.copy_(self.flat)
            buf = self.comm
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        dist.all_reduce(buf, op=op, group=self.group)
        if self.comm is not None:
            self.flat.copy_(self.comm)
        if not self._avg:
            self.flat.mul_(1.0 / self.world)

    def finish(self):
        """After backward: every bucket exchanged and visible to the current stream (no host synchronisation on GPU)."""
        while self._next < len(self.buckets):
            self._issue(self._next)
            self._next += 1
        if self._xs is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._xs)      # (no_comm / zero-filled buckets: no work to wait on)
        for k, w in self._works:
            w.wait()                   # nccl: the current stream waits for the communicator's stream; gloo: blocks
            if self.comm is not None:
                a, b, _ = self.buckets[k]
                self.flat[a:b].copy_(self.comm[a:b])
        if not self._avg and not self.no_comm:
            self.flat.mul_(1.0 / self.world)
        self._works = []
        self._reset()

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_state(modules, src=0, group=None):
    """Every rank starts from rank `src`'s parameters and buffers (what DDP's constructor does)."""
    _refuse_under_capture("broadcast_state")
    with torch.no_grad():
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t, src, group=group)


def mean_over_ranks(values, device, group=None):
    """Scalars (one per metric) averaged over the job: one all-reduce of len(values) numbers."""
    if dist.is_available() and dist.is_initialized():
        _refuse_under_capture("mean_over_ranks")          # (also with one rank: float() below would synchronise the capture)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(v) for v in values]
    dev = device if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    valid = torch.isfinite(t).to(torch.float64)
    t = torch.where(valid > 0, t, torch.zeros_like(t))
    both = torch.stack([t, valid])
    dist.all_reduce(both, group=group)
    return [float(s / c) if c > 0 else float("nan") for s, c in zip(both[0].tolist(), both[1].tolist())]
