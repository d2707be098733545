"""Data-parallel exchange step of the hot path.

Images shard across ranks with no data-path collective; the one exchange is the all-reduce(sum)/world of the
head-gradient bucket (the reference gets the same from apex DistributedDataParallel, detection/init.py:80-86).
One flat fp32 bucket per step -> one RCCL ring over xGMI instead of one small collective per parameter.
Works on any torch.distributed backend (``nccl`` = RCCL on ROCm; ``gloo`` in the CPU tests)."""
import torch
import torch.distributed as dist


class GradBucket(object):
    def __init__(self, params):
        self.params = [p for p in params]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def allreduce_(self, group=None, average=True):
        """In-place all-reduce of ``p.grad`` for every parameter of the bucket."""
        self.start_(group)
        self.finish_(group, average)

    def start_(self, group=None):
        """Packs the gradients into the flat bucket and STARTS the all-reduce (async): whatever the caller launches next
        -- the backward of the layers in front of these parameters -- overlaps with the ring.  ``finish_`` waits and
        writes the averaged gradients back."""
        self._work, self._grads, self._views = None, None, None
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError('GradBucket: a parameter has no gradient')
        dev = grads[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.empty((self.numel,), dtype=torch.float32, device=dev)
        views, off = [], 0
        for g in grads:
            views.append(self.flat[off:off + g.numel()].view(g.shape))
            off += g.numel()
        torch._foreach_copy_(views, grads)   # strided (channels_last) grads are laid into the bucket logically
        self._work = dist.all_reduce(self.flat, group=group, async_op=True)
        self._grads, self._views = grads, views

    def finish_(self, group=None, average=True):
        if getattr(self, '_work', None) is None:
            return
        self._work.wait()
        if average:
            self.flat.div_(dist.get_world_size(group))
        torch._foreach_copy_(self._grads, self._views)
        self._work, self._grads, self._views = None, None, None


def shard_batch(items, rank, world):
    """Contiguous shard of a list of per-image items for this rank (images are independent units)."""
    n = len(items)
    per = (n + world - 1) // world
    return items[rank * per:min(n, (rank + 1) * per)]
