"""Autograd wrappers over libssdk's generic NHWC convolution and BatchNorm (csrc/conv.hip, csrc/norm.hip).

Tensors are logical [B,C,H,W] with channels_last memory (= the NHWC buffers the kernels read); weights are logical
[Cout,Cin,k,k] with channels_last memory (= [Cout][k][k][Cin]).  GPU only -- CPU tensors raise.
"""
import torch

from . import _lib
from .distributed import grad_sink


def _nhwc(x):
    return x.float().contiguous(memory_format=torch.channels_last)


def _dp(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _out_dim(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def _fast_min_flops():
    """fast mode: smaller launches stay fp32 (the library applies the same bound, from the same variable, to the data gradients)."""
    e = __import__('os').environ.get('SSDK_FAST_MIN_FLOPS')
    return float(e) if e else 1.0e9


class _ConvFn(torch.autograd.Function):
    """apply(weight, bias, stride, pad, relu, stats, *xs) -> tuple of outputs; the n inputs share `weight` (one grouped launch).
    stats: None, or one fp64 `sums` buffer address (or None) per input: the BatchNorm statistics of that output are accumulated into it."""

    @staticmethod
    def forward(ctx, weight, bias, stride, pad, relu, stats, *xs):
        lib = _lib.lib()
        _lib.require_cuda(weight, *xs)
        w = weight.float().contiguous(memory_format=torch.channels_last)
        b = None if bias is None else bias.float().contiguous()
        cout, cin, k, k2 = w.shape
        assert k == k2
        xs = [_nhwc(x) for x in xs]
        B = xs[0].shape[0]
        ys = []
        arr = (_lib.ConvDesc * len(xs))()
        for i, x in enumerate(xs):
            assert x.shape[0] == B and x.shape[1] == cin, (tuple(x.shape), cin)
            ho, wo = _out_dim(x.shape[2], k, stride, pad), _out_dim(x.shape[3], k, stride, pad)
            y = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            ys.append(y)
            d = arr[i]
            d.x, d.hin, d.win, d.cin = _dp(x), x.shape[2], x.shape[3], cin
            d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = _dp(w), _dp(b), cout, k, stride, pad, int(bool(relu))
            d.y = _dp(y)
            d.stats = None if stats is None else stats[i]
        # opt-in split-bf16 forward (heads.set_fast_mode) -- for launches of at least ~1 GFLOP: below that the split of the weights and the
        # statistics pass behind the launch cost more than three bf16 MFMAs save over one fp32 one (the SSD tail's small layers)
        flops = 2.0 * B * cout * cin * k * k * sum(_out_dim(x.shape[2], k, stride, pad) * _out_dim(x.shape[3], k, stride, pad) for x in xs)
        if _lib.fast_mode == 'bf16x3' and cin % 32 == 0 and flops >= _fast_min_flops():
            fw = _lib.scratch(lib.ssdk_conv2d_fwd_fast_workspace_bytes(arr, len(xs)), xs[0].device, 'conv_fwd_fast')
            _lib.check(lib.ssdk_conv2d_fwd_fast(arr, len(xs), B, 3, _dp(fw), fw.numel(), _lib.current_stream()), 'ssdk_conv2d_fwd_fast')
        else:
            _conv2d_fwd(lib, arr, len(xs), B, xs[0].device)
        ctx.w_t = _transposed_weights_of(weight, stride)   # (prepare_weight_transposes ran for this step: the backward skips its re-layout)
        ctx.save_for_backward(w, *xs, *(ys if relu == 1 else []))
        # relu == 2: ReLU in the forward epilogue as usual, but its gradient is taken by the consumer (a BatchNorm that masks its dx where
        # its input is not positive, conv2d_batch_norm): the backward gets dy already masked and skips its own pass over it
        ctx.meta = (stride, pad, relu == 1, len(xs), bias is not None, tuple(tuple(y.shape) for y in ys))
        ctx.params = (weight, bias)   # leaves: their gradient-bucket slots (distributed.GradBucket) are looked up in the backward
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.lib()
        stride, pad, relu, n, has_bias, yshapes = ctx.meta
        saved = ctx.saved_tensors
        w, xs = saved[0], saved[1:1 + n]
        ys = saved[1 + n:] if relu else [None] * n
        cout, cin, k, _ = w.shape
        B = xs[0].shape[0]
        stream = _lib.current_stream()
        need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and has_bias
        # weight gradients of single-input convolutions can wait for the end of the backward pass: nothing in the chain depends on
        # them, and eight small launches that each underfill the chip become one grouped launch (_flush_weight_gradients)
        defer = (_defer_wgrad and n == 1 and (need_w or need_b) and ctx.params[0].is_leaf and (ctx.params[1] is None or ctx.params[1].is_leaf)
                 and not _has_hooks(ctx.params[0]) and not _has_hooks(ctx.params[1]))
        dw = db = None
        if need_w and not defer:   # zeroed by the library
            dw = grad_sink(ctx.params[0])
            if dw is None or dw.stride() != w.stride():
                dw = torch.empty_like(w, memory_format=torch.channels_last)
        if need_b and not defer:
            db = grad_sink(ctx.params[1])
            if db is None:
                db = torch.empty((cout,), dtype=torch.float32, device=w.device)
        arr = (_lib.ConvDesc * n)()
        dxs, keep = [], []
        for i in range(n):
            x = xs[i]
            dy = dys[i]   # an output nobody used downstream arrives as None
            dy = torch.zeros(yshapes[i], dtype=torch.float32, device=w.device).contiguous(memory_format=torch.channels_last) if dy is None else _nhwc(dy)
            if relu:  # undo the ReLU fused into the forward epilogue
                g = torch.empty_like(dy, memory_format=torch.channels_last)
                _lib.check(lib.ssdk_relu_bwd(_dp(ys[i]), _dp(dy), dy.numel(), _dp(g), stream), 'ssdk_relu_bwd')
                dy = g
            keep.append(dy)
            dx = torch.empty_like(x, memory_format=torch.channels_last) if ctx.needs_input_grad[6 + i] else None
            dxs.append(dx)
            d = arr[i]
            d.x, d.hin, d.win, d.cin = _dp(x), x.shape[2], x.shape[3], cin
            d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = _dp(w), None, cout, k, stride, pad, 0
            d.dy, d.dx, d.dw, d.db = _dp(dy), _dp(dx), _dp(dw), _dp(db)
            d.w_t = _dp(ctx.w_t)
        if defer:
            # jobs belong to ONE autograd graph task: a backward pass that raised leaves its callbacks unrun and its jobs behind -- they are
            # dropped (never added into a later step's gradients), and the flush is queued once per task, not "when the list was empty"
            _queue_deferred_wgrad(dict(x=xs[0], dy=keep[0], w=w, weight=ctx.params[0] if need_w else None, bias=ctx.params[1] if need_b else None,
                                       stride=stride, pad=pad))
            if all(dx is None for dx in dxs):
                return (None, None, None, None, None, None) + tuple(dxs)
        _conv2d_bwd(lib, arr, n, B, w.device, stream)
        _forget_transposed_weights(ctx.params[0])
        return (dw, db, None, None, None, None) + tuple(dxs)


# stream-K for the generic convolutions: the library takes it for launches of two rounds of tiles and more (conv.hip streamk_would_take:
# the RetinaNet tower, the large maps of the M2Det neck; slower on the one-round launches of a pyramid tail), so these calls carry the
# heads' 33 MB stream-K workspace (one cached buffer per device and stream); SSDK_CONV_STREAMK_GENERIC=0: never, no workspace.  Read once
_GENERIC_STREAMK = __import__('os').environ.get('SSDK_CONV_STREAMK_GENERIC', '') != '0'


def _conv2d_fwd(lib, arr, n, batch, device):
    if _GENERIC_STREAMK:
        sk = _lib.scratch(lib.ssdk_heads_fwd_workspace_bytes(), device, _lib.STREAMK_TAG, zeroed=True)   # (the stream-K state the heads use too)
        _lib.check(lib.ssdk_conv2d_fwd_ws(arr, n, batch, _dp(sk), sk.numel(), _lib.current_stream()), 'ssdk_conv2d_fwd')
    else:   # (the library's sticky stream-K error word is checked either way)
        _lib.check(lib.ssdk_conv2d_fwd_ws(arr, n, batch, None, 0, _lib.current_stream()), 'ssdk_conv2d_fwd')


def _conv2d_bwd(lib, arr, n, batch, device, stream):
    """ssdk_conv2d_bwd, or -- in the opt-in fast mode -- ssdk_conv2d_bwd_fast (stride-1 data gradients on the split-bf16 GEMM)."""
    if _lib.fast_mode == 'bf16x3':
        need = lib.ssdk_conv2d_bwd_fast_workspace_bytes(arr, n, batch)
        ws = _lib.scratch(need, device, 'conv2d_bwd')
        _lib.check(lib.ssdk_conv2d_bwd_fast(arr, n, batch, 0, 3, _dp(ws), ws.numel(), stream), 'ssdk_conv2d_bwd_fast')
    else:
        need = lib.ssdk_conv2d_bwd_workspace_bytes(arr, n, batch)
        ws = _lib.scratch(need, device, 'conv2d_bwd')
        if _GENERIC_STREAMK:   # (the stride-1 data gradients of a tower take stream-K like its forward launch: same state, same rule)
            sk = _lib.scratch(lib.ssdk_heads_fwd_workspace_bytes(), device, _lib.STREAMK_TAG, zeroed=True)
            _lib.check(lib.ssdk_conv2d_bwd_sk(arr, n, batch, 0, _dp(ws), ws.numel(), _dp(sk), sk.numel(), stream), 'ssdk_conv2d_bwd')
        else:
            _lib.check(lib.ssdk_conv2d_bwd(arr, n, batch, 0, _dp(ws), ws.numel(), stream), 'ssdk_conv2d_bwd')


class _GroupConvFn(torch.autograd.Function):
    """apply(meta, x_0, w_0, b_0, x_1, w_1, b_1, ...) -> outputs: n <= 8 INDEPENDENT convolutions -- own input, weights, kernel size, stride --
    in one grouped launch forward and one grouped call backward (the six scale branches of an M2Det TUM's smoothing layers, SFAM's per-scale
    gates: bf/modules/features.py:267, 290-296).  meta[i] = (stride, pad, relu, stats): relu / stats as in _ConvFn."""

    @staticmethod
    def forward(ctx, meta, *flat):
        lib = _lib.lib()
        n = len(meta)
        xs = [_nhwc(flat[3 * i]) for i in range(n)]
        weights = [flat[3 * i + 1] for i in range(n)]
        biases = [flat[3 * i + 2] for i in range(n)]
        _lib.require_cuda(*xs, *weights)
        ws = [w.float().contiguous(memory_format=torch.channels_last) for w in weights]
        bs = [None if b is None else b.float().contiguous() for b in biases]
        B = xs[0].shape[0]
        arr = (_lib.ConvDesc * n)()
        ys = []
        for i in range(n):
            stride, pad, relu, stats = meta[i]
            x, w = xs[i], ws[i]
            cout, cin, k, k2 = w.shape
            assert k == k2 and x.shape[0] == B and x.shape[1] == cin, (tuple(x.shape), tuple(w.shape))
            ho, wo = _out_dim(x.shape[2], k, stride, pad), _out_dim(x.shape[3], k, stride, pad)
            y = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            ys.append(y)
            d = arr[i]
            d.x, d.hin, d.win, d.cin = _dp(x), x.shape[2], x.shape[3], cin
            d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = _dp(w), _dp(bs[i]), cout, k, stride, pad, int(bool(relu))
            d.y = _dp(y)
            d.stats = stats
        _conv2d_fwd(lib, arr, n, B, xs[0].device)
        ctx.w_ts = [_transposed_weights_of(weights[i], meta[i][0]) for i in range(n)]
        ctx.save_for_backward(*ws, *xs, *[ys[i] if meta[i][2] == 1 else xs[i].new_empty(0) for i in range(n)])
        ctx.meta = tuple((m[0], m[1], m[2] == 1) for m in meta)
        ctx.params = [(weights[i], biases[i]) for i in range(n)]
        ctx.yshapes = [tuple(y.shape) for y in ys]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.lib()
        n = len(ctx.meta)
        saved = ctx.saved_tensors
        ws, xs, ysv = saved[:n], saved[n:2 * n], saved[2 * n:3 * n]
        B = xs[0].shape[0]
        stream = _lib.current_stream()
        arr = (_lib.ConvDesc * n)()
        out = [None]
        keep = []
        any_launch = False
        for i in range(n):
            stride, pad, relu = ctx.meta[i]
            w, x = ws[i], xs[i]
            weight, bias = ctx.params[i]
            cout, cin, k, _ = w.shape
            need_x, need_w, need_b = ctx.needs_input_grad[1 + 3 * i], ctx.needs_input_grad[2 + 3 * i], ctx.needs_input_grad[3 + 3 * i] and bias is not None
            dy = dys[i]
            dy = torch.zeros(ctx.yshapes[i], dtype=torch.float32, device=w.device).contiguous(memory_format=torch.channels_last) if dy is None else _nhwc(dy)
            if relu:
                g = torch.empty_like(dy, memory_format=torch.channels_last)
                _lib.check(lib.ssdk_relu_bwd(_dp(ysv[i]), _dp(dy), dy.numel(), _dp(g), stream), 'ssdk_relu_bwd')
                dy = g
            keep.append(dy)
            defer = (_defer_wgrad and (need_w or need_b) and weight.is_leaf and (bias is None or bias.is_leaf) and not _has_hooks(weight) and not _has_hooks(bias))
            dw = db = None
            if need_w and not defer:
                dw = grad_sink(weight)
                if dw is None or dw.stride() != w.stride():
                    dw = torch.empty_like(w, memory_format=torch.channels_last)
            if need_b and not defer:
                db = grad_sink(bias)
                if db is None:
                    db = torch.empty((cout,), dtype=torch.float32, device=w.device)
            dx = torch.empty_like(x, memory_format=torch.channels_last) if need_x else None
            d = arr[i]
            d.x, d.hin, d.win, d.cin = _dp(x), x.shape[2], x.shape[3], cin
            d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = _dp(w), None, cout, k, stride, pad, 0
            d.dy, d.dx, d.dw, d.db = _dp(dy), _dp(dx), _dp(dw), _dp(db)
            d.w_t = _dp(ctx.w_ts[i])
            any_launch = any_launch or dx is not None or dw is not None or db is not None
            if defer:
                _queue_deferred_wgrad(dict(x=x, dy=dy, w=w, weight=weight if need_w else None, bias=bias if need_b else None, stride=stride, pad=pad))
            out += [dx, dw, db]
        if any_launch:
            _conv2d_bwd(lib, arr, n, B, ws[0].device, stream)
        for weight, _ in ctx.params:
            _forget_transposed_weights(weight)
        return tuple(out)


def _queue_deferred_wgrad(job):
    """One deferred weight-gradient job of the current autograd graph task (see _ConvFn.backward)."""
    global _flush_queued_for
    task = torch._C._current_graph_task_id()
    if any(j['task'] != task for j in _pending_wgrads):
        _pending_wgrads[:] = [j for j in _pending_wgrads if j['task'] == task]
    job['task'] = task
    job['stream'] = torch.cuda.current_stream()   # (the node's stream: x and dy are complete there, the flush enqueues behind them)
    _pending_wgrads.append(job)
    if _flush_queued_for != task:
        _flush_queued_for = task
        torch.autograd.Variable._execution_engine.queue_callback(_flush_weight_gradients)


def conv2d_bn_group(xs, blocks):
    """[block(x) for block, x in zip(blocks, xs)] for INDEPENDENT Conv2dBn blocks (bf/modules/conv.py:30-36: conv -> BatchNorm -> ReLU) with
    the n <= 8 convolutions in ONE grouped launch (and one grouped backward call); the norms' statistics come from the convolutions'
    epilogues where they can.  Blocks the kernels do not take (see Conv2dBn._why_not_hip) must not be passed."""
    n = len(blocks)
    assert n == len(xs) and 0 < n <= 8
    chains = []
    for x, blk in zip(xs, blocks):
        bn = blk._modules.get('bn')
        ok = bn is not None and type(bn) is torch.nn.BatchNorm2d and x.is_cuda and torch.is_grad_enabled() and blk.conv.out_channels % 4 == 0
        chains.append(_local_training_chain(bn, x.device) if ok else None)
    global fused_stats_calls
    for c in chains:
        if c is not None:
            c.clean[0] = False
    fused_stats_calls += sum(c is not None for c in chains)
    meta, flat = [], []
    for x, blk, c in zip(xs, blocks, chains):
        cv = blk.conv
        has_bn, has_act = 'bn' in blk._modules, 'activation' in blk._modules
        meta.append((cv.stride[0], cv.padding[0], 1 if (has_act and not has_bn) else 0, None if c is None else c.buf[0].data_ptr()))
        flat += [x, cv.weight, cv.bias]
    ys = _GroupConvFn.apply(tuple(meta), *flat)
    out = []
    for y, blk, c in zip(ys, blocks, chains):
        bn = blk._modules.get('bn')
        has_act = 'activation' in blk._modules
        if bn is None:
            out.append(y)
        elif c is None:
            out.append(batch_norm(y, bn, relu=has_act) if type(bn) is torch.nn.BatchNorm2d else (torch.relu(bn(y)) if has_act else bn(y)))
        else:
            out.append(_BatchNormFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                                          True, int(has_act), c, True))
    return out


def conv2d_group(xs, convs, relu=False):
    """[conv(x) (-> ReLU)] for independent nn.Conv2d layers, one grouped launch (SFAM's per-scale fc1 / fc2, features.py:290-296)."""
    meta, flat = [], []
    for x, cv in zip(xs, convs):
        meta.append((cv.stride[0], cv.padding[0], 1 if relu else 0, None))
        flat += [x, cv.weight, cv.bias]
    return list(_GroupConvFn.apply(tuple(meta), *flat))


_defer_wgrad = False
_pending_wgrads = []
_flush_queued_for = None   # the autograd graph task (torch._C._current_graph_task_id) whose end-of-backward callback is queued


def _has_hooks(p):
    """Tensor hooks / post-accumulate-grad hooks on a parameter: they only fire when autograd itself delivers the gradient, so such a
    parameter's gradient is never deferred.  (The one exception: a parameter whose only hook is libssdk's own exchange wrapper --
    distributed.BucketedDataParallel marks it ``_ssdk_exchange_hook`` -- which flushes the deferred gradients itself before it starts
    the bucket they belong to.)"""
    if p is None:
        return False
    if getattr(p, '_backward_hooks', None):
        return True
    post = getattr(p, '_post_accumulate_grad_hooks', None)
    return bool(post) and not (len(post) == 1 and getattr(p, '_ssdk_exchange_hook', False))

# weights re-laid out for the backward-data GEMMs, keyed by id(parameter): (weakref to it, (version, data_ptr), stride == 1, the layout)
_wt_cache = {}


def _transposed_weights_of(weight, stride):
    ent = _wt_cache.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != (weight._version, weight.data_ptr()) or ent[2] != (stride == 1):
        return None   # (another tensor, changed in place, or its storage swapped through .data since the layout was made)
    return ent[3]


def _forget_all_transposed_weights(*_args, **_kwargs):
    """Global optimizer-step post hook: EVERY prepared layout is dropped when any optimizer has stepped.  An entry normally lives from
    ``prepare_weight_transposes`` to the backward pass of the same step, which forgets it; but a forward pass without a backward pass, or a
    backward pass that raised, leaves it behind, and a fused optimizer step changes the weights without bumping ``_version`` -- a later
    forward pass that does not prepare again would then multiply with the old step's weights (advisor, round 4)."""
    _wt_cache.clear()


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg
    _reg(_forget_all_transposed_weights)
    del _reg
except ImportError:   # (an older torch: the per-backward forgetting and the version / data_ptr key are what is left)
    pass


def _forget_transposed_weights(weight):
    """A prepared layout serves ONE backward pass (the one of the forward pass prepare_weight_transposes preceded): an optimizer step comes
    next, and a fused one leaves no trace in the parameter's version counter."""
    _wt_cache.pop(id(weight), None)


def prepare_weight_transposes(module):
    """Call at the start of a training step's forward pass over a chain of libssdk convolutions (the pyramid tail, the RetinaNet
    tower): the weights of every ``nn.Conv2d`` under ``module`` that the hot-path blocks run on libssdk are re-laid out for their
    backward-data GEMMs in ONE launch (``ssdk_conv2d_transpose_weights``) instead of one small launch in front of every backward call
    (8 per SSD-300 step, 40 per tower step: a shared tower weight was re-laid out once per level).  The layouts are keyed by parameter
    identity and version, so a stale one is never used: after an optimizer step (or for a convolution that was not prepared) the
    backward simply re-lays out its own weights as before.  No-op when gradients are off.  The entries only carry the layout from this
    call to the backward pass of the forward pass it precedes: the next call makes new ones."""
    import weakref
    import ctypes
    if not torch.is_grad_enabled():
        return 0
    convs = [m for m in module.modules() if isinstance(m, torch.nn.Conv2d) and m.groups == 1 and m.weight.is_cuda and m.weight.requires_grad
             and m.weight.dtype == torch.float32 and m.weight.is_contiguous(memory_format=torch.channels_last)
             and m.kernel_size[0] == m.kernel_size[1] and m.stride[0] == m.stride[1] and m.in_channels % 4 == 0 and m.out_channels % 4 == 0]
    # EVERY call re-lays out every weight: a layout made by an earlier step is never trusted.  (Round 3 skipped weights whose cache entry
    # still matched the parameter's ``_version`` -- but a fused optimizer step (torch.optim.SGD(fused=True): torch._fused_sgd_) updates
    # the parameter WITHOUT bumping its version counter, so from the second step on the backward-data GEMMs multiplied with the weights
    # of the step in which the entry was made.  Found by the deterministic-mode test, tools/determinism_step_diag.py.)
    seen, todo = set(), []
    for m in convs:
        w = m.weight
        if id(w) in seen:
            continue
        seen.add(id(w))
        todo.append((w, m.kernel_size[0], m.stride[0]))
    if not todo:
        return 0
    lib = _lib.lib()
    dev = todo[0][0].device
    sizes = [(w.numel() + 63) // 64 * 64 for w, _, _ in todo]   # (256-byte aligned slices of one arena)
    arena = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
    off = 0
    for first in range(0, len(todo), 24):
        group = todo[first:first + 24]
        arr = (_lib.ConvDesc * len(group))()
        outs = (ctypes.c_void_p * len(group))()
        for i, (w, k, stride) in enumerate(group):
            cout, cin = w.shape[0], w.shape[1]
            d = arr[i]
            d.w, d.cin, d.cout, d.ksize, d.stride = w.data_ptr(), cin, cout, k, stride
            view = arena[off:off + w.numel()]
            off += sizes[first + i]
            outs[i] = view.data_ptr()
            _wt_cache[id(w)] = (weakref.ref(w), (w._version, w.data_ptr()), stride == 1, view)
        _lib.check(lib.ssdk_conv2d_transpose_weights(arr, len(group), outs, _lib.current_stream()), 'ssdk_conv2d_transpose_weights')
    for key in [k_ for k_, v in _wt_cache.items() if v[0]() is None]:   # parameters that no longer exist
        del _wt_cache[key]
    return len(todo)


def defer_weight_gradients(enabled=True):
    """OPT-IN (off by default, process-wide).  Weight (and bias) gradients of single-input ``conv2d`` calls are computed at the END of the
    backward pass, all in grouped launches (up to eight convolutions each), and written into ``param.grad`` directly -- autograd's
    AccumulateGrad never sees them.  Restrictions: ``torch.autograd.grad`` with respect to such a weight returns None for it (and the
    flush still adds into ``.grad``); do NOT enable it under a wrapper that waits for AccumulateGrad (torch DistributedDataParallel);
    parameters with tensor hooks / post-accumulate-grad hooks are never deferred.  libssdk's own GradBucket path is fine (the flush runs
    before ``backward()`` returns).  A backward pass that raises leaves nothing behind: its jobs are dropped by the next pass.
    Returns the previous setting."""
    global _defer_wgrad
    prev = _defer_wgrad
    _defer_wgrad = bool(enabled)
    return prev


def set_deterministic(enabled=True):
    """Process-wide deterministic mode of libssdk (``ssdk_set_deterministic``; also ``SSDK_DETERMINISTIC=1`` in the environment) -- the
    counterpart of the reference's ``torch.backends.cudnn.deterministic = True`` (bf/training/env.py:74-76).  On: no fp32 atomics anywhere
    in the training kernels -- convolutions are not split over K, data gradients take the output-stationary form (the heads' sparse
    scatter forms are not used), weight and bias gradients are reduced in a fixed order -- so the same inputs give the same bits, run to
    run and eager vs HIP-graph replay.  Slower (bench.py reports ``deterministic_ms_per_step`` per config).  Returns the previous
    setting; ``with ops.deterministic():`` scopes it."""
    return bool(_lib.lib().ssdk_set_deterministic(1 if enabled else 0))


def is_deterministic():
    return bool(_lib.lib().ssdk_get_deterministic())


class deterministic(object):
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = set_deterministic(self.enabled)
        return self

    def __exit__(self, *exc):
        set_deterministic(self.prev)
        return False


class deferred_weight_gradients(object):
    """``with ops.deferred_weight_gradients():`` -- the opt-in of ``defer_weight_gradients`` for the backward passes run inside the block
    only (the previous setting comes back on exit, also when the block raises): a training step that owns its parameters' gradients
    (bench.HotPath.train_step) scopes it to itself instead of changing what ``torch.autograd.grad`` returns for the rest of the process."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = defer_weight_gradients(self.enabled)
        return self

    def __exit__(self, *exc):
        defer_weight_gradients(self.prev)
        return False


def _flush_weight_gradients():
    global _flush_queued_for
    task = torch._C._current_graph_task_id()
    jobs = [j for j in _pending_wgrads if j['task'] == task or task < 0]
    del _pending_wgrads[:]   # (jobs of another task are leftovers of a backward pass that raised)
    _flush_queued_for = None
    if not jobs:
        return
    # jobs queued by nodes of another stream (a pyramid tail differentiated beside the heads: detection/modules/heads.py
    # multi_level_heads_split) are flushed on THAT stream, behind the launches that made their operands, and the caller's stream --
    # on which the optimizer runs next -- waits for it
    caller = torch.cuda.current_stream()
    streams = []
    for j in jobs:
        if all(j['stream'] != s_ for s_ in streams):
            streams.append(j['stream'])
    for s_ in streams:
        mine = [j for j in jobs if j['stream'] == s_]
        if s_ == caller:
            _flush_jobs(mine)
        else:
            with torch.cuda.stream(s_):
                _flush_jobs(mine)
            caller.wait_stream(s_)


def _flush_jobs(jobs):
    lib = _lib.lib()
    stream = _lib.current_stream()
    for first in range(0, len(jobs), 8):
        group = jobs[first:first + 8]
        if len({j['x'].shape[0] for j in group}) != 1:
            group_list = [[j] for j in group]   # (a grouped call shares the batch size)
        else:
            group_list = [group]
        for grp in group_list:
            arr = (_lib.ConvDesc * len(grp))()
            outs = []
            for i, j in enumerate(grp):
                w, x = j['w'], j['x']
                cout, cin, k, _ = w.shape
                dw = db = None
                if j['weight'] is not None:
                    dw = grad_sink(j['weight'])
                    if dw is None or dw.stride() != w.stride():
                        dw = torch.empty_like(w, memory_format=torch.channels_last)
                if j['bias'] is not None:
                    db = grad_sink(j['bias'])
                    if db is None:
                        db = torch.empty((cout,), dtype=torch.float32, device=w.device)
                d = arr[i]
                d.x, d.hin, d.win, d.cin = _dp(x), x.shape[2], x.shape[3], cin
                d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = _dp(w), None, cout, k, j['stride'], j['pad'], 0
                d.dy, d.dx, d.dw, d.db = _dp(j['dy']), None, _dp(dw), _dp(db)
                outs.append((j['weight'], dw, j['bias'], db))
            B = grp[0]['x'].shape[0]
            need = lib.ssdk_conv2d_bwd_workspace_bytes(arr, len(grp), B)
            ws = _lib.scratch(need, grp[0]['w'].device, 'conv2d_bwd')
            _lib.check(lib.ssdk_conv2d_bwd(arr, len(grp), B, 0, _dp(ws), ws.numel(), stream), 'ssdk_conv2d_bwd')
            for weight, dw, bias, db in outs:
                for p_, g_ in ((weight, dw), (bias, db)):
                    if p_ is None:
                        continue
                    if p_.grad is None:
                        p_.grad = g_
                    else:
                        p_.grad = p_.grad + g_


def conv2d(xs, weight, bias=None, stride=1, padding=0, relu=False):
    """Conv2d on a list of maps that share the weights (one grouped GEMM launch); returns a list."""
    single = isinstance(xs, torch.Tensor)
    out = _ConvFn.apply(weight, bias, int(stride), int(padding), bool(relu), None, *([xs] if single else list(xs)))
    return out[0] if single else list(out)


_NO_FUSED_STATS = bool(__import__('os').environ.get('SSDK_NO_FUSED_STATS'))   # (measurement knob)
_NO_BN_CHAIN = bool(__import__('os').environ.get('SSDK_NO_BN_CHAIN'))         # (measurement knob: every norm call zero-fills a scratch workspace)
_NO_RELU_BY_NORM = bool(__import__('os').environ.get('SSDK_NO_RELU_BY_NORM'))   # (measurement knob)
fused_stats_calls = 0   # (norm layers whose forward statistics were handed to a convolution's epilogue; tests read it)


def _local_training_chain(bn, device):
    """The layer's `sums` chain when its forward statistics can be taken by the producing convolution's epilogue: a plain BatchNorm2d in
    training mode, per-process statistics, its forward buffer known to hold zeros.  None otherwise."""
    if _NO_FUSED_STATS or type(bn) is not torch.nn.BatchNorm2d or not bn.training or bn.momentum is None or not bn.track_running_stats or sync_group_of(bn) is not None:
        return None
    chain = getattr(bn, '_ssdk_sums_chain', None)
    if chain is None or chain.buf.device != device or chain.buf.shape[1] != 2 * bn.num_features + 2:
        chain = bn._ssdk_sums_chain = _SumsChain(bn.num_features, device)
    return chain if chain.usable(0, device) else None


def conv2d_batch_norm(xs, weight, bias, stride, padding, bns, conv_relu=False, bn_relu=False):
    """conv (-> ReLU) -> BatchNorm (-> ReLU) on a list of maps that share the convolution's weights and have a norm layer each
    (bf/modules/conv.py:30-36; the per-level norms of a RetinaNet tower layer, detection/modules/predictors.py:60-76).  Where a norm is a
    training-mode BatchNorm2d with per-process statistics, the convolution's epilogue accumulates its statistics (ssdk_conv_desc::stats)
    and the norm is ONE launch (apply) instead of two -- and the statistics pass over the activation is gone; anywhere else this is
    conv2d followed by batch_norm."""
    single = isinstance(xs, torch.Tensor)
    xs = [xs] if single else list(xs)
    bns = [bns] if single else list(bns)
    chains = [_local_training_chain(bn, x.device) if x.is_cuda and torch.is_grad_enabled() else None for x, bn in zip(xs, bns)]
    if not any(c is not None for c in chains) or weight.shape[0] % 4:
        ys = conv2d(xs, weight, bias, stride, padding, relu=conv_relu)
        out = [batch_norm(y, bn, relu=bn_relu) for y, bn in zip(ys, bns)]
        return out[0] if single else out
    for c in chains:
        if c is not None:
            c.clean[0] = False   # (handed to the convolution: whatever happens next, it no longer holds zeros)
    global fused_stats_calls
    fused_stats_calls += sum(c is not None for c in chains)
    stats = tuple(None if c is None else c.buf[0].data_ptr() for c in chains)
    # conv -> ReLU -> norm with every norm on this path: the norms' backward also takes the ReLU's gradient (dx = 0 where the norm's input
    # is not positive), and the convolution's backward skips its own pass over dy
    relu_by_norm = bool(conv_relu) and all(c is not None for c in chains) and not _NO_RELU_BY_NORM
    ys = _ConvFn.apply(weight, bias, int(stride), int(padding), 2 if relu_by_norm else int(bool(conv_relu)), stats, *xs)
    out = []
    for y, bn, c in zip(ys, bns, chains):
        if c is None:
            out.append(batch_norm(y, bn, relu=bn_relu))
        else:
            out.append(_BatchNormFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                                          True, int(bool(bn_relu)) | (2 if relu_by_norm else 0), c, True))
    return out[0] if single else out


class _SumsChain(object):
    """Two fp64 `sums` buffers of one BatchNorm layer (forward, backward) and which of them currently hold zeros.  A forward call
    accumulates into the forward buffer and has its apply launch zero the backward one; the backward call does the reverse: in
    steady-state training no zero-fill launch runs for the layer at all (they were 16 of the ~150 launches of an SSD-300 step).
    Whenever the buffer a call needs is not known to be clean -- a second forward before the backward, an evaluation pass in training
    mode, another stream -- the call takes the plain entry point, which zero-fills a scratch workspace itself."""

    def __init__(self, channels, device):
        self.buf = torch.zeros((2, 2 * channels + 2), dtype=torch.float64, device=device)
        self.clean = [True, True]   # forward buffer, backward buffer
        self.stream = _lib.raw_stream(device)

    def usable(self, which, device):
        return not _NO_BN_CHAIN and self.clean[which] and self.buf.device == device and self.stream == _lib.raw_stream(device)


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, relu, chain, sums_ready=False):
        lib = _lib.lib()
        _lib.require_cuda(x)
        x = _nhwc(x)
        B, C, H, W = x.shape
        rows = B * H * W
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.float().contiguous()
        b = None if beta is None else beta.float().contiguous()
        relu = int(relu)      # bit 0: the norm's own fused ReLU; bit 1 (backward only): the input is a ReLU output, its gradient is taken here too
        own = relu & 1
        if sums_ready:   # the producing convolution's epilogue left the statistics in chain.buf[0] (conv2d_batch_norm)
            _lib.check(lib.ssdk_batchnorm_apply_chained(_dp(x), rows, C, _dp(g), _dp(b), _dp(running_mean), _dp(running_var), _dp(num_batches_tracked),
                                                        float(momentum), float(eps), own, _dp(y), _dp(mean), _dp(rstd), chain.buf[0].data_ptr(),
                                                        chain.buf[1].data_ptr(), _lib.current_stream()), 'ssdk_batchnorm_apply_chained')
            chain.clean = [False, True]
        elif training and chain is not None and chain.usable(0, x.device):
            _lib.check(lib.ssdk_batchnorm_fwd_chained(_dp(x), rows, C, _dp(g), _dp(b), _dp(running_mean), _dp(running_var), _dp(num_batches_tracked),
                                                      float(momentum), float(eps), own, _dp(y), _dp(mean), _dp(rstd), chain.buf[0].data_ptr(),
                                                      chain.buf[1].data_ptr(), _lib.current_stream()), 'ssdk_batchnorm_fwd_chained')
            chain.clean = [False, True]
        else:
            ws = _lib.scratch(lib.ssdk_batchnorm_workspace_bytes(C), x.device, 'batchnorm')
            _lib.check(lib.ssdk_batchnorm_fwd(_dp(x), rows, C, _dp(g), _dp(b), _dp(running_mean), _dp(running_var),
                                              _dp(num_batches_tracked) if training else None, float(momentum), float(eps), int(training), own, _dp(y), _dp(mean), _dp(rstd), _dp(ws), ws.numel(),
                                              _lib.current_stream()), 'ssdk_batchnorm_fwd')
        ctx.save_for_backward(x, y if own else x.new_empty(0), g if g is not None else x.new_empty(0), mean, rstd)
        ctx.meta = (relu, bool(training), gamma is not None, beta is not None)
        ctx.chain = chain
        ctx.mark_non_differentiable(mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x, y, g, mean, rstd = ctx.saved_tensors
        relu, training, has_g, has_b = ctx.meta
        chain = ctx.chain
        B, C, H, W = x.shape
        dy = _nhwc(dy)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dgamma = torch.empty((C,), dtype=torch.float32, device=x.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=x.device)
        if training and chain is not None and chain.usable(1, x.device):
            _lib.check(lib.ssdk_batchnorm_bwd_chained(_dp(x), _dp(y) if relu & 1 else None, _dp(dy), B * H * W, C, _dp(g) if has_g else None, _dp(mean),
                                                      _dp(rstd), relu, _dp(dx), _dp(dgamma), _dp(dbeta), chain.buf[1].data_ptr(),
                                                      chain.buf[0].data_ptr(), _lib.current_stream()), 'ssdk_batchnorm_bwd_chained')
            chain.clean = [True, False]
        else:
            ws = _lib.scratch(lib.ssdk_batchnorm_workspace_bytes(C), x.device, 'batchnorm')
            _lib.check(lib.ssdk_batchnorm_bwd(_dp(x), _dp(y) if relu & 1 else None, _dp(dy), B * H * W, C, _dp(g) if has_g else None, _dp(mean),
                                              _dp(rstd), relu, int(training), _dp(dx), _dp(dgamma), _dp(dbeta), _dp(ws), ws.numel(),
                                              _lib.current_stream()), 'ssdk_batchnorm_bwd')
        return dx, dgamma if has_g else None, dbeta if has_b else None, None, None, None, None, None, None, None, None, None


def batch_norm(x, bn, relu=False):
    """``bn`` is a torch.nn.BatchNorm2d (its parameters / buffers are used and updated exactly like torch does)."""
    if bn.momentum is None or not bn.track_running_stats:
        raise NotImplementedError('BatchNorm2d with momentum=None / track_running_stats=False is not on the GPU path')
    training = bn.training
    mark = sync_group_of(bn)
    if mark is not None and training:   # statistics over all ranks (detection.init(distributed=True))
        return sync_batch_norm([x], [bn], relu, mark[0])[0]
    nbt = bn.num_batches_tracked   # incremented inside the library's launch (torch: a launch of its own per layer)
    if nbt is not None:
        assert nbt.dtype == torch.int64 and nbt.is_cuda
    chain = None
    if training and x.is_cuda:
        chain = getattr(bn, '_ssdk_sums_chain', None)
        if chain is None or chain.buf.device != x.device or chain.buf.shape[1] != 2 * bn.num_features + 2:
            chain = bn._ssdk_sums_chain = _SumsChain(bn.num_features, x.device)
    return _BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, nbt, bn.momentum, bn.eps, training, relu, chain, False)


def sync_group_of(bn):
    """The process group a BatchNorm2d was marked with by distributed.convert_sync_batchnorm (detection.init(distributed=True)), as a
    1-tuple -- (None,) is the default group -- or None for a plain, per-process BatchNorm."""
    return getattr(bn, '_ssdk_sync_group', None)


def allreduce_sums_(buf, group=None):
    """The one exchange step of synchronised BatchNorm: all-reduce(sum) of a packed fp64 buffer [sum_i (2 C_i + 2)] holding, per
    norm layer, (sum, sum of squares / products, row count, padding).  No-op without an initialised process group or with one rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, group=group)
    return buf


class _SyncBatchNormFn(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ fused ReLU) over the GLOBAL batch for n layers at once (e.g. the five per-level norms of one
    RetinaNet tower layer): per-layer partial sums on libssdk -> ONE all-reduce of the packed buffer -> per-layer apply.
    apply(group, relu, n, x_0, gamma_0, beta_0, running_mean_0, running_var_0, nbt_0, momentum_0, eps_0, x_1, ...) -> y_0, y_1, ...
    (apex convert_syncbn_model in the reference, detection/init.py:85)."""

    @staticmethod
    def forward(ctx, group, relu, n, *args):
        lib = _lib.lib()
        stream = _lib.current_stream()
        layers = []
        for i in range(n):
            x, gamma, beta, rm, rv, nbt, momentum, eps = args[8 * i:8 * i + 8]
            _lib.require_cuda(x)
            x = _nhwc(x)
            layers.append(dict(x=x, C=x.shape[1], rows=x.shape[0] * x.shape[2] * x.shape[3],
                               g=None if gamma is None else gamma.float().contiguous(), b=None if beta is None else beta.float().contiguous(),
                               rm=rm, rv=rv, nbt=nbt, momentum=float(momentum), eps=float(eps)))
        dev = layers[0]['x'].device
        offs, tot = [], 0
        for lv in layers:
            offs.append(tot)
            tot += 2 * lv['C'] + 2   # (sum, sum of squares, rows, padding)
        sums = torch.empty((tot,), dtype=torch.float64, device=dev)
        for lv, off in zip(layers, offs):
            _lib.check(lib.ssdk_batchnorm_stats(_dp(lv['x']), lv['rows'], lv['C'], sums[off:].data_ptr(), stream), 'ssdk_batchnorm_stats')
        allreduce_sums_(sums, group)
        ys, saved = [], []
        for lv, off in zip(layers, offs):
            x, C = lv['x'], lv['C']
            y = torch.empty_like(x, memory_format=torch.channels_last)
            mean = torch.empty((C,), dtype=torch.float32, device=dev)
            rstd = torch.empty((C,), dtype=torch.float32, device=dev)
            _lib.check(lib.ssdk_batchnorm_apply(_dp(x), lv['rows'], C, _dp(lv['g']), _dp(lv['b']), _dp(lv['rm']), _dp(lv['rv']), _dp(lv['nbt']),
                                                lv['momentum'], lv['eps'], int(relu), _dp(y), _dp(mean), _dp(rstd), sums[off:].data_ptr(), 1, stream),
                       'ssdk_batchnorm_apply')
            ys.append(y)
            saved += [x, y if relu else x.new_empty(0), lv['g'] if lv['g'] is not None else x.new_empty(0), mean, rstd]
        ctx.save_for_backward(*saved)
        ctx.meta = (group, bool(relu), n, [(lv['C'], lv['rows'], lv['g'] is not None, lv['b'] is not None) for lv in layers], offs, tot)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.lib()
        stream = _lib.current_stream()
        group, relu, n, shapes, offs, tot = ctx.meta
        saved = ctx.saved_tensors
        dev = saved[0].device
        sums = torch.empty((tot,), dtype=torch.float64, device=dev)
        dys = [torch.zeros_like(saved[5 * i], memory_format=torch.channels_last) if dy is None else _nhwc(dy) for i, dy in enumerate(dys)]
        for i in range(n):
            x, y, g, mean, rstd = saved[5 * i:5 * i + 5]
            C, rows, _, _ = shapes[i]
            _lib.check(lib.ssdk_batchnorm_bwd_stats(_dp(x), _dp(y) if relu else None, _dp(dys[i]), rows, C, _dp(mean), _dp(rstd), int(relu),
                                                    sums[offs[i]:].data_ptr(), stream), 'ssdk_batchnorm_bwd_stats')
        local = sums.clone()   # the affine parameters' gradients are this rank's own sums (the gradient exchange averages them later)
        allreduce_sums_(sums, group)
        out = [None, None, None]
        for i in range(n):
            x, y, g, mean, rstd = saved[5 * i:5 * i + 5]
            C, rows, has_g, has_b = shapes[i]
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
            base = sums[offs[i]:].data_ptr()
            _lib.check(lib.ssdk_batchnorm_bwd_apply(_dp(x), _dp(y) if relu else None, _dp(dys[i]), rows, C, _dp(g) if has_g else None, _dp(mean),
                                                    _dp(rstd), int(relu), 1, base, local[offs[i]:].data_ptr(), base + 16 * C, _dp(dx), _dp(dgamma),
                                                    _dp(dbeta), stream), 'ssdk_batchnorm_bwd_apply')
            out += [dx, dgamma if has_g else None, dbeta if has_b else None, None, None, None, None, None]
        return tuple(out)


def sync_batch_norm(xs, bns, relu=False, group=None):
    """Synchronised training-mode BatchNorm of several maps with their own norm layers, one exchange for all of them."""
    args = []
    for x, bn in zip(xs, bns):
        if bn.momentum is None or not bn.track_running_stats:
            raise NotImplementedError('BatchNorm2d with momentum=None / track_running_stats=False is not on the GPU path')
        args += [x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps]
    return list(_SyncBatchNormFn.apply(group, bool(relu), len(bns), *args))


def batch_norm_levels(xs, bns, relu=False):
    """One BatchNorm2d per map (the per-level norms of a RetinaNet tower layer, detection/modules/predictors.py:39-42,69-70).  Layers
    marked for synchronisation share ONE packed all-reduce in training mode; otherwise each runs on its own."""
    marks = [sync_group_of(bn) for bn in bns]
    if all(m is not None for m in marks) and all(bn.training for bn in bns) and len({id(m[0]) for m in marks}) == 1:
        return sync_batch_norm(xs, bns, relu, marks[0][0])
    return [batch_norm(x, bn, relu) for x, bn in zip(xs, bns)]


class _UpsampleAddFn(torch.autograd.Function):
    """out = fine + nearest_upsample(coarse, size of fine)   (FPN top-down step, bf/modules/features.py:106-107)"""

    @staticmethod
    def forward(ctx, fine, coarse):
        _lib.require_cuda(fine, coarse)
        fine, coarse = _nhwc(fine), _nhwc(coarse)
        B, C, Hf, Wf = fine.shape
        assert coarse.shape[0] == B and coarse.shape[1] == C
        out = torch.empty_like(fine, memory_format=torch.channels_last)
        _lib.check(_lib.lib().ssdk_upsample_nearest_add_fwd(_dp(fine), _dp(coarse), B, Hf, Wf, coarse.shape[2], coarse.shape[3], C, _dp(out),
                                                            _lib.current_stream()), 'ssdk_upsample_nearest_add_fwd')
        ctx.shapes = (B, C, Hf, Wf, coarse.shape[2], coarse.shape[3])
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, Hf, Wf, Hc, Wc = ctx.shapes
        dout = _nhwc(dout)
        dcoarse = None
        if ctx.needs_input_grad[1]:
            dcoarse = torch.empty((B, C, Hc, Wc), dtype=torch.float32, device=dout.device, memory_format=torch.channels_last)
            _lib.check(_lib.lib().ssdk_upsample_nearest_add_bwd(_dp(dout), B, Hf, Wf, Hc, Wc, C, _dp(dcoarse), _lib.current_stream()),
                       'ssdk_upsample_nearest_add_bwd')
        return (dout if ctx.needs_input_grad[0] else None), dcoarse


def upsample_add(fine, coarse):
    return _UpsampleAddFn.apply(fine, coarse)


class _UpsampleFn(torch.autograd.Function):
    """F.interpolate(x, size=(h, w), mode='nearest') (bf/modules/features.py:371)."""

    @staticmethod
    def forward(ctx, coarse, hf, wf):
        _lib.require_cuda(coarse)
        coarse = _nhwc(coarse)
        B, C, Hc, Wc = coarse.shape
        out = torch.empty((B, C, hf, wf), dtype=torch.float32, device=coarse.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().ssdk_upsample_nearest_add_fwd(None, _dp(coarse), B, hf, wf, Hc, Wc, C, _dp(out), _lib.current_stream()),
                   'ssdk_upsample_nearest_add_fwd')
        ctx.shapes = (B, C, hf, wf, Hc, Wc)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, Hf, Wf, Hc, Wc = ctx.shapes
        dout = _nhwc(dout)
        dcoarse = torch.empty((B, C, Hc, Wc), dtype=torch.float32, device=dout.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().ssdk_upsample_nearest_add_bwd(_dp(dout), B, Hf, Wf, Hc, Wc, C, _dp(dcoarse), _lib.current_stream()),
                   'ssdk_upsample_nearest_add_bwd')
        return dcoarse, None, None


def upsample_nearest(x, size):
    return _UpsampleFn.apply(x, int(size[0]), int(size[1]))


class _AvgPoolFn(torch.autograd.Function):
    """F.adaptive_avg_pool2d(x, 1) -> [B, C, 1, 1]"""

    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        x = _nhwc(x)
        B, C, H, W = x.shape
        out = torch.empty((B, C, 1, 1), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ssdk_global_avgpool_fwd(_dp(x), B, H * W, C, _dp(out), _lib.current_stream()), 'ssdk_global_avgpool_fwd')
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, H, W = ctx.shape
        dout = dout.float().contiguous()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dout.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().ssdk_global_avgpool_bwd(_dp(dout), B, H * W, C, _dp(dx), _lib.current_stream()), 'ssdk_global_avgpool_bwd')
        return dx


def global_avg_pool(x):
    return _AvgPoolFn.apply(x)


class _GateFn(torch.autograd.Function):
    """x * sigmoid(z), z [B, C, 1, 1] (bf/modules/features.py:296-298)"""

    @staticmethod
    def forward(ctx, x, z):
        _lib.require_cuda(x, z)
        x = _nhwc(x)
        z = z.float().contiguous()
        B, C, H, W = x.shape
        out = torch.empty_like(x, memory_format=torch.channels_last)
        _lib.check(_lib.lib().ssdk_sigmoid_gate_fwd(_dp(x), _dp(z), B, H * W, C, _dp(out), _lib.current_stream()), 'ssdk_sigmoid_gate_fwd')
        ctx.save_for_backward(x, z)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, z = ctx.saved_tensors
        B, C, H, W = x.shape
        dout = _nhwc(dout)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dz = torch.empty_like(z)
        _lib.check(_lib.lib().ssdk_sigmoid_gate_bwd(_dp(x), _dp(z), _dp(dout), B, H * W, C, _dp(dx), _dp(dz), _lib.current_stream()),
                   'ssdk_sigmoid_gate_bwd')
        return dx, dz


def sigmoid_gate(x, z):
    return _GateFn.apply(x, z)


class _SfamFn(torch.autograd.Function):
    """M2Det SFAM over PIECES: apply(n_scales, n_pieces, *pieces (scale-major), *(fc1.weight, fc1.bias, fc2.weight, fc2.bias per scale))
    -> n_scales gated maps [B, n_pieces * Cp, H_s, W_s].  What bf/modules/features.py:385 + :286-298 compute -- torch.cat of a scale's TUM
    outputs, adaptive_avg_pool2d, fc1 + ReLU, fc2, sigmoid gate -- without ever building the concatenated map: the pool and the gate read
    the pieces where they are (ssdk_sfam_*), the two 1 x 1 "fc" convolutions of all scales run as grouped launches, and the backward pass
    writes each piece's gradient (gate's part + pool's part) as its own contiguous map in one pass."""

    @staticmethod
    def forward(ctx, n_scales, n_pieces, *flat):
        import ctypes
        lib = _lib.lib()
        stream = _lib.current_stream()
        pieces = [[_nhwc(flat[s * n_pieces + k]) for k in range(n_pieces)] for s in range(n_scales)]
        params = flat[n_scales * n_pieces:]
        _lib.require_cuda(*[p for ps in pieces for p in ps])
        w1 = [params[4 * s].float().contiguous(memory_format=torch.channels_last) for s in range(n_scales)]
        b1 = [None if params[4 * s + 1] is None else params[4 * s + 1].float().contiguous() for s in range(n_scales)]
        w2 = [params[4 * s + 2].float().contiguous(memory_format=torch.channels_last) for s in range(n_scales)]
        b2 = [None if params[4 * s + 3] is None else params[4 * s + 3].float().contiguous() for s in range(n_scales)]
        B, Cp = pieces[0][0].shape[0], pieces[0][0].shape[1]
        C, Hc = n_pieces * Cp, w1[0].shape[0]
        dev = pieces[0][0].device
        for s in range(n_scales):
            shape = pieces[s][0].shape
            assert all(p.shape == shape for p in pieces[s]) and shape[0] == B and shape[1] == Cp, 'the pieces of a scale must share one shape'
            assert tuple(w1[s].shape) == (Hc, C, 1, 1) and tuple(w2[s].shape) == (C, Hc, 1, 1), (tuple(w1[s].shape), tuple(w2[s].shape))
        pooled = torch.empty((n_scales, B, C), dtype=torch.float32, device=dev)
        hidden = torch.empty((n_scales, B, Hc), dtype=torch.float32, device=dev)
        z = torch.empty((n_scales, B, C), dtype=torch.float32, device=dev)
        ptrs = [(ctypes.c_void_p * n_pieces)(*[p.data_ptr() for p in pieces[s]]) for s in range(n_scales)]
        hw = [pieces[s][0].shape[2] * pieces[s][0].shape[3] for s in range(n_scales)]
        for s in range(n_scales):
            _lib.check(lib.ssdk_sfam_pool_fwd(ptrs[s], n_pieces, B, hw[s], Cp, pooled[s].data_ptr(), stream), 'ssdk_sfam_pool_fwd')
        for x, w, b, y, relu in ((pooled, w1, b1, hidden, 1), (hidden, w2, b2, z, 0)):   # fc1 + F.relu, fc2: one grouped launch each
            arr = (_lib.ConvDesc * n_scales)()
            for s in range(n_scales):
                d = arr[s]
                d.x, d.hin, d.win, d.cin = x[s].data_ptr(), 1, 1, x.shape[2]
                d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = w[s].data_ptr(), _dp(b[s]), y.shape[2], 1, 1, 0, relu
                d.y = y[s].data_ptr()
            _conv2d_fwd(lib, arr, n_scales, B, dev)
        outs = []
        for s in range(n_scales):
            out = torch.empty((B, C) + tuple(pieces[s][0].shape[2:]), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
            _lib.check(lib.ssdk_sfam_gate_fwd(ptrs[s], n_pieces, B, hw[s], Cp, z[s].data_ptr(), _dp(out), stream), 'ssdk_sfam_gate_fwd')
            outs.append(out)
        ctx.save_for_backward(pooled, hidden, z, *w1, *w2, *[p for ps in pieces for p in ps])
        ctx.meta = (n_scales, n_pieces, [b is not None for b in b1], [b is not None for b in b2])
        ctx.params = params
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        import ctypes
        lib = _lib.lib()
        stream = _lib.current_stream()
        n_scales, n_pieces, has_b1, has_b2 = ctx.meta
        saved = ctx.saved_tensors
        pooled, hidden, z = saved[:3]
        w1, w2 = saved[3:3 + n_scales], saved[3 + n_scales:3 + 2 * n_scales]
        flat_pieces = saved[3 + 2 * n_scales:]
        pieces = [flat_pieces[s * n_pieces:(s + 1) * n_pieces] for s in range(n_scales)]
        B, Cp = pieces[0][0].shape[0], pieces[0][0].shape[1]
        C, Hc = n_pieces * Cp, hidden.shape[2]
        dev = pooled.device
        hw = [pieces[s][0].shape[2] * pieces[s][0].shape[3] for s in range(n_scales)]
        ptrs = [(ctypes.c_void_p * n_pieces)(*[p.data_ptr() for p in pieces[s]]) for s in range(n_scales)]
        douts = [torch.zeros((B, C) + tuple(pieces[s][0].shape[2:]), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
                 if g is None else _nhwc(g) for s, g in enumerate(douts)]
        # 1. dz = sigmoid'(z) * sum_hw dout * piece
        dz = torch.empty_like(z)
        for s in range(n_scales):
            _lib.check(lib.ssdk_sfam_gate_bwd_reduce(ptrs[s], n_pieces, B, hw[s], Cp, z[s].data_ptr(), _dp(douts[s]), dz[s].data_ptr(), stream),
                       'ssdk_sfam_gate_bwd_reduce')
        # 2. fc2 and fc1 backward (data, weight and bias gradients; one grouped call each), the ReLU between them
        d_hidden = torch.empty_like(hidden)
        dpool = torch.empty_like(pooled)
        dw1 = [torch.empty_like(w, memory_format=torch.channels_last) for w in w1]
        dw2 = [torch.empty_like(w, memory_format=torch.channels_last) for w in w2]
        db1 = [torch.empty((Hc,), dtype=torch.float32, device=dev) if h else None for h in has_b1]
        db2 = [torch.empty((C,), dtype=torch.float32, device=dev) if h else None for h in has_b2]

        def conv_bwd(x, w, dy, dx, dw, db):
            arr = (_lib.ConvDesc * n_scales)()
            for s in range(n_scales):
                d = arr[s]
                d.x, d.hin, d.win, d.cin = x[s].data_ptr(), 1, 1, x.shape[2]
                d.w, d.bias, d.cout, d.ksize, d.stride, d.pad, d.relu = w[s].data_ptr(), None, dy.shape[2], 1, 1, 0, 0
                d.dy, d.dx, d.dw, d.db = dy[s].data_ptr(), dx[s].data_ptr(), dw[s].data_ptr(), _dp(db[s])
            _conv2d_bwd(lib, arr, n_scales, B, dev, stream)
        conv_bwd(hidden, w2, dz, d_hidden, dw2, db2)
        g_hidden = torch.empty_like(d_hidden)
        _lib.check(lib.ssdk_relu_bwd(_dp(hidden), _dp(d_hidden), hidden.numel(), _dp(g_hidden), stream), 'ssdk_relu_bwd')
        conv_bwd(pooled, w1, g_hidden, dpool, dw1, db1)
        # 3. every piece's gradient: dout * sigmoid(z) + dpool / HW, one pass
        dpieces = []
        for s in range(n_scales):
            ds = [torch.empty_like(p, memory_format=torch.channels_last) for p in pieces[s]]
            darr = (ctypes.c_void_p * n_pieces)(*[t.data_ptr() for t in ds])
            _lib.check(lib.ssdk_sfam_gate_bwd_apply(darr, n_pieces, B, hw[s], Cp, z[s].data_ptr(), _dp(douts[s]), dpool[s].data_ptr(), stream),
                       'ssdk_sfam_gate_bwd_apply')
            dpieces += ds
        need = ctx.needs_input_grad
        grads = [g if need[2 + i] else None for i, g in enumerate(dpieces)]
        base = 2 + n_scales * n_pieces
        for s in range(n_scales):
            for j, g in enumerate((dw1[s], db1[s], dw2[s], db2[s])):
                grads.append(g if need[base + 4 * s + j] else None)
        return (None, None) + tuple(grads)


def sfam_pieces(pieces, fc1, fc2):
    """[cat(pieces_s, 1) * sigmoid(fc2_s(relu(fc1_s(avgpool(cat(pieces_s, 1)))))) for every scale s] (bf/modules/features.py:385, :286-298)
    with no concatenated map (``_SfamFn``).  ``pieces``: per scale the list of [B, Cp, H_s, W_s] maps; ``fc1`` / ``fc2``: per scale the 1 x 1
    nn.Conv2d layers.  ``sfam_pieces_ok`` says whether the kernels take the arguments."""
    n_scales, n_pieces = len(pieces), len(pieces[0])
    flat = [p for ps in pieces for p in ps]
    for a, b in zip(fc1, fc2):
        flat += [a.weight, a.bias, b.weight, b.bias]
    return list(_SfamFn.apply(n_scales, n_pieces, *flat))


def sfam_pieces_ok(pieces, fc1, fc2):
    if not pieces or len(pieces) != len(fc1) or len(pieces) != len(fc2) or len(pieces) > 8:
        return False
    n = len(pieces[0])
    if not (0 < n <= 8) or any(len(ps) != n for ps in pieces):
        return False
    first = pieces[0][0]
    Cp, B = first.shape[1], first.shape[0]
    for ps, a, b in zip(pieces, fc1, fc2):
        if any((not p.is_cuda) or p.dim() != 4 or p.dtype != torch.float32 or p.shape != ps[0].shape or p.shape[1] != Cp or p.shape[0] != B for p in ps):
            return False
        for m in (a, b):
            if not isinstance(m, torch.nn.Conv2d) or m.kernel_size != (1, 1) or m.stride != (1, 1) or m.padding != (0, 0) or m.groups != 1:
                return False
        if a.in_channels != n * Cp or b.out_channels != n * Cp or a.out_channels != b.in_channels or a.out_channels != fc1[0].out_channels:
            return False
    return Cp % 4 == 0 and fc1[0].out_channels % 4 == 0 and (B * fc1[0].out_channels * len(pieces)) % 4 == 0


class _DepthwiseFn(torch.autograd.Function):
    """Depthwise k x k convolution (groups = channels) on libssdk; weight is torch's [C, 1, k, k] parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        lib = _lib.lib()
        _lib.require_cuda(x, weight)
        x = _nhwc(x)
        B, C, H, W = x.shape
        k = weight.shape[-1]
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        ho, wo = _out_dim(H, k, stride, pad), _out_dim(W, k, stride, pad)
        y = torch.empty((B, C, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        _lib.check(lib.ssdk_depthwise_conv2d_fwd(_dp(x), _dp(w), _dp(b), B, H, W, C, k, stride, pad, _dp(y), _lib.current_stream()),
                   'ssdk_depthwise_conv2d_fwd')
        ctx.save_for_backward(x, w)
        ctx.meta = (stride, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x, w = ctx.saved_tensors
        stride, pad, has_bias = ctx.meta
        B, C, H, W = x.shape
        k = w.shape[-1]
        dy = _nhwc(dy)
        dx = torch.empty_like(x, memory_format=torch.channels_last) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty((C,), dtype=torch.float32, device=x.device) if has_bias else None
        _lib.check(lib.ssdk_depthwise_conv2d_bwd(_dp(x), _dp(w), _dp(dy), B, H, W, C, k, stride, pad, _dp(dx), _dp(dw), _dp(db), 0,
                                                 _lib.current_stream()), 'ssdk_depthwise_conv2d_bwd')
        return dx, dw, db, None, None


def depthwise_conv2d(x, weight, bias=None, stride=1, padding=0):
    return _DepthwiseFn.apply(x, weight, bias, int(stride), int(padding))
