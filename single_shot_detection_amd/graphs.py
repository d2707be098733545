"""HIP graphs for launch-bound steps.

At serving batch sizes the evaluation step (pyramid tail + heads forward + postprocess: ~25 launches of a few microseconds each) is bound
by the host's enqueue rate, not by the GPU: SSD-300 at batch 1 takes 0.60 ms enqueued launch by launch and 0.46 ms replayed from a graph
(batch 2: 0.75 -> 0.50 ms).  ``GraphedCallable`` captures any function of device tensors once (``torch.cuda.CUDAGraph`` = hipGraph on
ROCm) and replays it on new inputs.

What makes libssdk's entry points capturable: they only enqueue kernels on the caller's stream, every workspace is owned by the caller
(``_lib.scratch`` buffers -- keyed by stream -- are created during the warm-up calls and the capture runs on that SAME side stream, so no
workspace is allocated, and no zero-fill of one recorded, inside the graph; ``scratch_allocated_in_capture`` counts violations), nothing is
read back to the host, and the stream-K
flags of ``ssdk_heads_fwd`` are reset by their consumer, so a replay (same launch arguments, same epoch) never sees the previous replay's
flags.  What the captured function itself must respect: fixed shapes, no ``.item()`` / ``.cpu()`` / host-side branching on device values
(``Postprocessor.postprocess_padded`` returns padded rows + counts for exactly this reason; ``postprocess`` splits on the host and is not
capturable), and its outputs are static buffers that the next call overwrites.

The training step can be captured as well (forward, match, sampler, loss, backward, fused SGD: ``GraphedCallable(step, [])``) once the
ground truth is a ``target_assigner.PackedGroundTruth`` -- static device buffers of fixed capacity refilled between replays -- instead of a
list of host tensors packed and copied inside the step: ssd_mb2_voc at batch 2 goes from 0.78 ms (host bound) to 0.27 ms per step; at
batch 32 the SSD-300 step is GPU bound and gains nothing.  Replays were checked against eager steps parameter by parameter
(tests/test_end_to_end_gpu.py).
"""
import torch


class GraphedCallable(object):
    def __init__(self, fn, example_args, warmup=3):
        self.static_in = [a.clone() for a in example_args]
        current = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(current)
        with torch.cuda.stream(side):   # (warm-up off the default stream, as CUDAGraph capture requires: allocates every scratch buffer)
            for _ in range(warmup):
                fn(*self.static_in)
        current.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        from . import _lib
        before = set(_lib._scratch)
        # capture on the warm-up stream: _lib.scratch and the BatchNorm `sums` chains are per stream, so what the warm-up created is what
        # the captured calls find (captured on another stream every workspace would be allocated again from the graph's private pool --
        # and the 33 MB stream-K workspace's zero-fill would be replayed with every step)
        with torch.cuda.graph(self.graph, stream=side):
            self.static_out = fn(*self.static_in)
        self.scratch_allocated_in_capture = len(set(_lib._scratch) - before)
        self.stream = side

    def __call__(self, *args):
        assert len(args) == len(self.static_in)
        from . import _lib
        if _lib.streamk_poisoned():   # (a replay never passes through ssdk_heads_fwd's own check; the kernels store NaN from then on)
            raise _lib.SsdkError('GraphedCallable: an earlier stream-K head GEMM of this process gave up waiting for a parked partial tile; '
                                 'its outputs (and every replay since) are NaN -- see _lib.streamk_timeouts()')
        for dst, src in zip(self.static_in, args):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError('GraphedCallable: captured for %s %s, called with %s %s' % (tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_out


def graphed_eval(detector, postprocessor, example_images, warmup=3):
    """detector(images) -> postprocessor.postprocess_padded, captured for ``example_images``' shape.  Returns a callable
    images -> (rows [B, max_total, 6], counts [B]) whose results live in static buffers (copy them before the next call if needed)."""
    detector.eval()

    def step(images):
        with torch.no_grad():
            scores, locs, priors = detector(images)
            return postprocessor.postprocess_padded((scores, locs), priors)
    return GraphedCallable(step, [example_images], warmup=warmup)



class GraphedSegment(object):
    """A differentiable segment of a training step -- ``fn(*inputs) -> (loss, *aux)`` -- captured as TWO HIP graphs (forward; backward of
    ``loss`` with respect to ``inputs`` and ``params``) and exposed as one autograd node, so that what sits in front of it (a PyTorch
    backbone) stays eager while the segment costs the host two graph launches instead of ~90 kernel launches.

    ``detection.init(..., graph_hot_path=True)`` puts the libssdk part of a training step in it: pyramid tail + heads forward, target
    assignment, sampler + multibox loss, and their backward (detection/init.py:108-135 between the backbone and ``loss.backward()``).

    * ``inputs``: example tensors (shapes are fixed); the call copies the real ones into static buffers (the backbone's taps: one device
      copy per step), the gradients with respect to them are returned to autograd as the static buffers themselves.
    * ``params``: their gradients are NOT routed through autograd: after the backward replay ``p.grad`` is set to the static gradient
      buffer (or the buffer is added to an existing ``p.grad``) -- no copy, but the tensor is overwritten by the next step's backward,
      and parameter hooks do not fire (not for ``distributed=True``).
    * only ``loss`` (output 0) is differentiable; outputs are static buffers the next call overwrites.
    * anything else the segment reads (ground truth, anchors) must live in static device buffers the caller refills between calls
      (``target_assigner.PackedGroundTruth``)."""

    def __init__(self, fn, inputs, params, warmup=2, defer_weight_gradients=True):
        from . import _lib, ops
        self.fn = fn
        self.params = [p for p in params if p.requires_grad]
        self.static_in = [x.detach().clone().requires_grad_(True) for x in inputs]
        # ``defer_weight_gradients``: the weight gradients of the segment's single-input convolutions (the pyramid tail) are computed at the
        # end of the captured backward pass in grouped launches (ops.defer_weight_gradients: 8 launches that each underfill the chip -> 1);
        # they reach ``param.grad`` directly, which is where this class picks the static buffers up
        held = [p.grad for p in self.params]
        for p in self.params:
            p.grad = None
        current = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(current)
        with torch.cuda.stream(side), ops.deferred_weight_gradients(bool(defer_weight_gradients)):
            # warm-up on the capture stream: every per-stream workspace exists before the capture (see GraphedCallable)
            for _ in range(warmup):
                outs = fn(*self.static_in)
                torch.autograd.grad(outs[0], self.static_in + self.params, allow_unused=True)
                for p in self.params:
                    p.grad = None
            del outs
        current.wait_stream(side)
        before = set(_lib._scratch)
        self.g_fwd, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, stream=side):
            self.static_out = tuple(fn(*self.static_in))
        self.static_gloss = torch.ones_like(self.static_out[0])
        with torch.cuda.graph(self.g_bwd, stream=side, pool=self.g_fwd.pool()), ops.deferred_weight_gradients(bool(defer_weight_gradients)):
            grads = torch.autograd.grad(self.static_out[0], self.static_in + self.params, grad_outputs=self.static_gloss, allow_unused=True)
        self.static_gin = grads[:len(self.static_in)]
        gparam = list(grads[len(self.static_in):])
        for i, p in enumerate(self.params):   # deferred gradients: written straight into a fresh p.grad by the captured flush
            if gparam[i] is None and p.grad is not None:
                gparam[i] = p.grad
            p.grad = held[i]
        self.static_gparam = tuple(gparam)
        self.scratch_allocated_in_capture = len(set(_lib._scratch) - before)
        self.stream = side
        self._anchor = torch.zeros((1,), device=self.static_in[0].device if self.static_in else 'cuda', requires_grad=True)   # (the node exists even when no input requires a gradient)
        self.collect_timing = False   # bench.py: events around the two replays (incl. the input copies / the gradient hand-over)
        self._ev = None

    def read_timing(self):
        """(forward ms, backward ms) of the last call with ``collect_timing`` on; synchronises with its events."""
        if not self._ev or len(self._ev) < 4:
            return None
        self._ev[3].synchronize()
        return self._ev[0].elapsed_time(self._ev[1]), self._ev[2].elapsed_time(self._ev[3])

    def __call__(self, *inputs):
        assert len(inputs) == len(self.static_in)
        return _SegmentFn.apply(self, self._anchor, *inputs)


class _SegmentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, anchor, *inputs):
        from . import _lib
        if _lib.streamk_poisoned():
            raise _lib.SsdkError('GraphedSegment: an earlier stream-K head GEMM of this process gave up waiting for a parked partial tile -- see _lib.streamk_timeouts()')
        if seg.collect_timing:
            seg._ev = [torch.cuda.Event(enable_timing=True)]
            seg._ev[0].record()
        for dst, src in zip(seg.static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError('GraphedSegment: captured for %s %s, called with %s %s' % (tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
            dst.detach().copy_(src, non_blocking=True)
        seg.g_fwd.replay()
        if seg.collect_timing:
            seg._ev.append(torch.cuda.Event(enable_timing=True))
            seg._ev[1].record()
        ctx.seg = seg
        ctx.n_in = len(inputs)
        outs = tuple(o.detach() for o in seg.static_out)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, gloss, *_unused):
        seg = ctx.seg
        timed = seg.collect_timing and seg._ev is not None and len(seg._ev) == 2
        if timed:
            seg._ev.append(torch.cuda.Event(enable_timing=True))
            seg._ev[2].record()
        seg.static_gloss.copy_(gloss.reshape(seg.static_gloss.shape), non_blocking=True)
        seg.g_bwd.replay()
        if timed:   # (before the host-side hand-over below: it launches nothing unless a gradient is accumulated)
            seg._ev.append(torch.cuda.Event(enable_timing=True))
            seg._ev[3].record()
        for p, g in zip(seg.params, seg.static_gparam):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g.detach()
            else:
                p.grad.add_(g)
        return (None, None) + tuple(None if g is None else g.detach() for g in seg.static_gin)
