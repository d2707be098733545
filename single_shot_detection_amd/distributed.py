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
        self._work = dist.all_reduce(self.flat, op=op, group=group, async_op=True)

    def finish_(self, group=None):
        """Waits for the collective ``start_`` began (the average was chosen there: ReduceOp.AVG, or a division here on gloo)."""
        if self._work is None:
            return
        self._work.wait()
        if self._post_div:
            self.flat.div_(self._post_div)
        self._work = None


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
