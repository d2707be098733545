"""Data-parallel exchange step of the hot path.

Images shard across ranks with no data-path collective; the one exchange is the all-reduce(sum)/world of the
head-gradient bucket (the reference gets the same from apex DistributedDataParallel, detection/init.py:80-86).
One flat fp32 bucket per step -> one RCCL ring over xGMI instead of one small collective per parameter.
Works on any torch.distributed backend (``nccl`` = RCCL on ROCm; ``gloo`` in the CPU tests)."""
import torch
import torch.distributed as dist


def grad_sink(param):
    """A fresh alias of the bucket slot GradBucket.attach_() reserved for ``param`` (None when there is none, or when the parameter
    already holds a gradient that this pass must be ADDED to).  Backward code that produces the parameter's whole gradient in one
    launch (heads / conv weight gradients) writes it there and returns the alias, so that autograd's AccumulateGrad adopts it as
    ``param.grad`` without a copy (it clones a gradient somebody else still references: hence a new alias per call).
    Contract of an attached parameter: one autograd node per backward pass produces its gradient (true of every head / extras /
    tower parameter: weights shared across pyramid levels are summed inside one grouped library call)."""
    view = getattr(param, '_ssdk_grad_view', None) if param is not None else None
    if view is None or param.grad is not None:
        return None
    return view.detach()


_avg_ok = {}


def _avg_supported(group, device):
    """ReduceOp.AVG exists in NCCL >= 2.10 / RCCL; gloo has no such op.  Probed once per (backend, group) with a one-element
    all-reduce instead of assumed: an unsupported op must cost a division per step, not the run."""
    key = (dist.get_backend(group), id(group))
    if key not in _avg_ok:
        ok = False
        if key[0] == 'nccl':
            try:
                t = torch.full((1,), float(dist.get_rank(group) + 1), dtype=torch.float32, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                n = dist.get_world_size(group)
                ok = abs(float(t.item()) - (n + 1) / 2.0) < 1e-4
            except Exception:
                ok = False
        _avg_ok[key] = ok
    return _avg_ok[key]


class GradBucket(object):
    """The gradients of ``params`` as ONE flat fp32 buffer: one collective per bucket instead of one per parameter.

    ``attach_()`` gives every parameter a slot (a strided view with the parameter's own memory layout) and publishes it as
    ``param._ssdk_grad_view``; producers that write there make ``param.grad`` a view of the bucket, and then ``start_`` / ``finish_``
    move no data at all (a gradient found elsewhere is copied in and the parameter's ``.grad`` re-pointed at its slot).
    The average is taken by the collective itself (ReduceOp.AVG) where the backend has it (nccl = RCCL); gloo sums, then one div_."""

    def __init__(self, params):
        self.params = [p for p in params]
        # every slot starts on a 256-byte boundary: fused optimizer kernels only vectorise 16-byte aligned tensors (unaligned slots
        # made the fused SGD step three times slower), and the collective moves whole lines anyway
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        self.flat = None
        self.views = None
        self._work = None
        # timing of the last collective (BucketedDataParallel.collect_timing): device events on the stream the collective was started /
        # joined on, and the host time finish_() blocked (what a host-synchronous backend -- gloo -- costs instead)
        self.timing = False
        self._ev = None
        self.ring_ms = None       # start_ -> joined (an UPPER bound of the ring's own duration: it includes whatever the stream still had queued)
        self.exposed_ms = None    # how long the joining stream (or, on gloo, the host) stalled for the ring

    @property
    def nbytes(self):
        return 4 * self.numel

    def attach_(self, device=None):
        dev = device if device is not None else self.params[0].device
        if self.flat is not None and self.flat.device == torch.device(dev):
            return self
        self.flat = torch.zeros((self.numel,), dtype=torch.float32, device=dev)
        self.views = []
        for p, off in zip(self.params, self.offsets):
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            v = torch.as_strided(self.flat, tuple(p.shape), p.stride() if dense else torch.empty(p.shape).stride(), off)
            self.views.append(v)
            p._ssdk_grad_view = v
        return self

    def allreduce_(self, group=None, average=True, force=False):
        """In-place all-reduce of ``p.grad`` for every parameter of the bucket."""
        self.start_(group, average, force)
        self.finish_(group)

    def start_(self, group=None, average=True, force=False):
        """STARTS the all-reduce (async): whatever the caller launches next -- the backward of the layers in front of these
        parameters -- overlaps with the ring.  ``finish_`` waits for it.  ``force``: run the collective with one rank too (a group of
        one leaves the values unchanged; tests use it to put the bucket through RCCL on a one-GPU box)."""
        self._work = None
        if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
            return
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError('GradBucket: a parameter has no gradient')
        self.attach_(grads[0].device)
        stray = [(v, g, p) for v, g, p in zip(self.views, grads, self.params)
                 if g.data_ptr() != v.data_ptr() or g.stride() != v.stride()]
        if stray:   # gradients that were not produced in place (small ones: BatchNorm affine, biases): one batched copy
            torch._foreach_copy_([v for v, _, _ in stray], [g for _, g, _ in stray])
            for v, _, p in stray:
                p.grad = v.detach()
        self.copied_last = len(stray)
        world = dist.get_world_size(group)
        self._post_div = None
        op = dist.ReduceOp.SUM
        if average:
            if _avg_supported(group, self.flat.device):
                op = dist.ReduceOp.AVG
            else:
                self._post_div = world
        if self.timing and self.flat.is_cuda:
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            self._ev[0].record()
        self._work = dist.all_reduce(self.flat, op=op, group=group, async_op=True)

    def finish_(self, group=None):
        """Waits for the collective ``start_`` began (the average was chosen there: ReduceOp.AVG, or a division here on gloo)."""
        if self._work is None:
            return
        import time
        t0 = time.perf_counter()
        if self._ev is not None:
            self._ev[1].record()
        self._work.wait()       # (nccl: the current stream waits, the host does not; gloo: the host waits)
        if self._ev is not None:
            self._ev[2].record()
        self._host_wait_ms = (time.perf_counter() - t0) * 1e3
        if self._post_div:
            self.flat.div_(self._post_div)
        self._work = None

    def read_timing_(self):
        """ring_ms / exposed_ms of the last collective (synchronises with its events; call outside the timed region)."""
        if self._ev is not None:
            self._ev[2].synchronize()
            self.ring_ms = self._ev[0].elapsed_time(self._ev[2])
            self.exposed_ms = self._ev[1].elapsed_time(self._ev[2])
            self._ev = None
        elif self.timing and getattr(self, '_host_wait_ms', None) is not None:
            self.ring_ms = None
            self.exposed_ms = self._host_wait_ms
        return self.ring_ms, self.exposed_ms


def shard_batch(items, rank, world):
    """Contiguous shard of a list of per-image items for this rank (images are independent units)."""
    n = len(items)
    per = (n + world - 1) // world
    return items[rank * per:min(n, (rank + 1) * per)]


def convert_sync_batchnorm(module, process_group=None):
    """The role of apex `convert_syncbn_model` (detection/init.py:85): every BatchNorm of the detector gets batch statistics over all
    ranks.  BatchNorm2d layers inside the hot-path blocks (Conv2dBn / DepthwiseConv2dBn pyramid tail and necks, the RetinaNet tower's
    per-level norms) stay nn.BatchNorm2d -- same parameters, same state_dict keys -- and are MARKED: libssdk computes their partial
    sums, torch.distributed all-reduces one packed buffer per layer (per tower layer across the levels), libssdk applies.  Everything
    else (the PyTorch backbone) is converted to torch.nn.SyncBatchNorm."""
    import torch.nn as nn
    from .bf.modules.conv import Conv2dBn, DepthwiseConv2dBn
    from .detection.modules.predictors import SharedConvPredictor
    hot = (Conv2dBn, DepthwiseConv2dBn, SharedConvPredictor)

    def walk(m, in_hot):
        for name, child in list(m.named_children()):
            h = in_hot or isinstance(child, hot)
            if isinstance(child, nn.modules.batchnorm._BatchNorm):
                if h and type(child) is nn.BatchNorm2d:
                    child._ssdk_sync_group = (process_group,)
                else:
                    setattr(m, name, nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group))
            else:
                walk(child, h)
    walk(module, isinstance(module, hot))
    return module


class BucketedDataParallel(torch.nn.Module):
    """One process per GPU, gradients averaged over the ranks through flat fp32 ``GradBucket``s -- the role of apex
    DistributedDataParallel in the reference (detection/init.py:80-86), and the exchange ``bench.py --gpus N`` measures.

    ``groups``: lists of parameters in the order their gradients complete during the backward pass; default: the heads' parameters
    (``module.heads``; complete as soon as the heads' backward node has run, i.e. first) and everything else.  A
    post-accumulate-grad hook per parameter counts arrivals; when a bucket is complete AND every bucket in front of it has started, its
    all-reduce STARTS (async) -- collectives are issued in bucket-index order on every rank, whatever order the gradients arrive in (ranks
    whose unused parameters differ would otherwise pair different buckets' rings) --, so the heads' ring runs under the backward pass of
    the pyramid tail / towers / backbone; a callback at the end of the backward pass starts whatever has not started (parameters that took
    no part in the step count as zero gradients: unlike apex / torch DDP, which leave such a ``.grad`` None, the optimizer then sees a zero
    gradient -- weight decay and momentum still move the parameter) and waits for all rings, so ``backward()`` returns with averaged
    gradients like DDP's does.  A backward pass that RAISES after some rings have started leaves this rank's collective count short of its
    peers'; the next pass (or ``abort_step_()``) first issues the missing collectives, so that the ranks stay paired -- the gradients of
    the failed step are meaningless, but nothing hangs and nothing pairs a 36 MB ring with a 9 MB one.  Producers that write a parameter's gradient into its bucket slot (``grad_sink``: the
    heads' / convolutions' weight-gradient kernels) make the exchange zero-copy.  ``no_sync()`` skips the exchange (gradient
    accumulation).  Parameters and buffers are broadcast from rank 0 at construction.  ``state_dict`` keys carry the ``module.`` prefix
    like DDP's."""

    def __init__(self, module, groups=None, process_group=None, broadcast=True):
        super(BucketedDataParallel, self).__init__()
        self.module = module
        self.process_group = process_group
        params = [p for p in module.parameters() if p.requires_grad]
        if groups is None:
            heads = getattr(module, 'heads', None)
            first = [p for p in heads.parameters() if p.requires_grad] if heads is not None else []
            ids = {id(p) for p in first}
            groups = [g for g in (first, [p for p in params if id(p) not in ids]) if g]
        seen = set()
        for g in groups:
            for p in g:
                assert id(p) not in seen, 'BucketedDataParallel: a parameter appears in two groups'
                seen.add(id(p))
        assert seen == {id(p) for p in params}, 'BucketedDataParallel: the groups must cover every parameter that requires a gradient'
        if broadcast and dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
        self.buckets = [GradBucket(g) for g in groups]
        for b in self.buckets:
            b.attach_(b.params[0].device)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b.params}
        self._arrived = [0] * len(self.buckets)
        self._started = [False] * len(self.buckets)
        self._callback_queued = False
        self._task = None              # the autograd graph task the queued callback belongs to
        self.require_sync = True
        self.start_order = []          # (diagnostics / tests: bucket indices in the order their rings started in the last backward pass)
        self.started_early = []        # ... and which of them started from a hook, i.e. before the backward pass had ended
        self.recovered_steps = 0       # backward passes that raised and whose missing collectives were issued afterwards
        self.collect_timing = False    # bench.py: event-timed ring duration / exposed wait per bucket (exchange_timing())
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for p in b.params]
        for b in self.buckets:
            for p in b.params:
                p._ssdk_exchange_hook = True   # (ops.defer_weight_gradients may still defer this parameter: _finish_all flushes first)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    class _NoSync(object):
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.prev, self.owner.require_sync = self.owner.require_sync, False

        def __exit__(self, *exc):
            self.owner.require_sync = self.prev
            return False

    def no_sync(self):
        return BucketedDataParallel._NoSync(self)

    def _active(self):
        return self.require_sync and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1

    def _on_grad(self, param):
        if not self._active():
            return
        task = torch._C._current_graph_task_id()
        if self._callback_queued and task != self._task:
            # the pass that queued the callback never reached it (its backward raised): its arrival counts and "started" marks are
            # not this pass's -- without this the next pass would neither queue a callback nor finish its rings (stale gradients, silently)
            self.abort_step_()
        if not self._callback_queued:
            self._callback_queued = True
            self._task = task
            self.start_order, self.started_early = [], []
            torch.autograd.Variable._execution_engine.queue_callback(self._finish_all)
        i = self._bucket_of[id(param)]
        self._arrived[i] += 1
        # strictly in index order: bucket i starts from a hook only once every bucket in front of it has started; a bucket that completes
        # out of order waits for its predecessors (or for the end of the pass)
        j = 0
        while j < len(self.buckets) and self._started[j]:
            j += 1
        while j < len(self.buckets) and self._arrived[j] == len(self.buckets[j].params):
            self._start(j, early=True)
            j += 1

    def abort_step_(self):
        """After a backward pass that raised: issue the collectives that pass never reached (with whatever the buckets hold), wait for all of
        them, reset the counters.  The peers issued theirs; this keeps the per-rank collective count -- and the pairing of the rings -- equal.
        Called by the next pass' first gradient hook; a trainer that catches the exception may call it itself."""
        if not self._callback_queued:
            return
        # (parameters without a gradient get a zero one for the owed collective and lose it again afterwards: the pass that follows must
        # not accumulate into the failed step's leftovers)
        none_before = [p for i, b in enumerate(self.buckets) if not self._started[i] for p in b.params if p.grad is None]
        try:
            for i in range(len(self.buckets)):
                if not self._started[i]:
                    self._start(i, early=False)
        finally:
            try:
                for b in self.buckets:
                    b.finish_(self.process_group)
            finally:
                for p in none_before:
                    p.grad = None
                self._arrived = [0] * len(self.buckets)
                self._started = [False] * len(self.buckets)
                self._callback_queued = False
                self.recovered_steps += 1

    def _start(self, i, early):
        b = self.buckets[i]
        for p, v in zip(b.params, b.views):
            if p.grad is None:   # took no part in this step: a zero gradient on every rank that agrees, the others' share otherwise
                p.grad = v.zero_().detach()
        b.timing = self.collect_timing
        b.start_(self.process_group)
        self._started[i] = True
        self.start_order.append(i)
        if early:
            self.started_early.append(i)

    def _finish_all(self):
        # Whatever raises on the way, every bucket's collective is issued and waited for before the counters go back: a rank that skipped
        # one would leave its peers waiting in it (and pair its next ring with their previous one), and a ring that is not waited for may
        # still be writing the buffer when the next start_ overwrites its handle.  The first error is raised at the end.
        err = None
        try:
            # weight gradients deferred to the end of the backward pass (ops.defer_weight_gradients) are written now, before the buckets
            # they belong to start: this callback was queued by the FIRST gradient of the pass, the deferred flush's own callback later
            from . import ops
            ops._flush_weight_gradients()
        except BaseException as e:   # noqa: B902
            err = e
        for i in range(len(self.buckets)):
            if not self._started[i]:
                try:
                    self._start(i, early=False)
                except BaseException as e:   # noqa: B902
                    err = err or e
        for b in self.buckets:
            try:
                b.finish_(self.process_group)
            except BaseException as e:   # noqa: B902
                err = err or e
        self._arrived = [0] * len(self.buckets)
        self._started = [False] * len(self.buckets)
        self._callback_queued = False
        if err is not None:
            raise err

    def exchange_timing(self):
        """Per bucket of the last exchanged step (collect_timing = True): {'ring_ms': start -> joined on the device timeline (upper bound of
        the ring), 'exposed_ms': how long the join stalled the stream (gloo: the host)}.  Synchronises with the step's events."""
        out = []
        for b in self.buckets:
            ring, exposed = b.read_timing_()
            out.append({'bytes': b.nbytes, 'ring_ms': ring, 'exposed_ms': exposed})
        return out
