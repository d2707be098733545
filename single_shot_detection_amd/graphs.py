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

