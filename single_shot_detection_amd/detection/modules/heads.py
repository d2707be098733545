"""Multi-scale head convolutions on libssdk (csrc/conv.hip) -- the arithmetic of detection/detector.py:50-66.

One autograd function covers every pyramid level: it launches one implicit-GEMM per level that writes directly
into the concatenated ``scores [B, A*C]`` / ``locs [B, A*4]`` buffers (no permute / contiguous / cat), and its
backward produces the source-map gradients, the weight gradients and the bias gradients per level.
"""
import torch

from ... import _lib
from ...distributed import grad_sink


def set_fast_mode(mode):
    """Opt-in reduced-precision mode of the forward GEMMs (the reference's analogue: apex AMP O1, bf/training/env.py:87-95).
    ``None`` (default): exact fp32 on v_mfma_f32_32x32x2_f32.  ``'bf16x3'``: operands split into bf16 pieces, three cross terms per product
    on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the head GEMM (ssdk_heads_fwd_fast) and every ``ops.conv2d`` whose input has
    Cin % 32 == 0 (ssdk_conv2d_fwd_fast: RetinaNet's towers, the pyramid tail); outputs within ~1e-5 of their scale; the backward pass is
    unchanged.  The heads need Cin % 32 == 0 on every level (the call raises otherwise); other convolutions stay fp32 where it does not
    hold.  Returns the previous setting."""
    if mode not in (None, 'bf16x3'):
        raise ValueError(f'fast mode {mode!r}: None or "bf16x3"')
    prev, _lib.fast_mode = _lib.fast_mode, mode
    return prev


# Measurement hook (bench.py): a list that receives one (start, end) pair of torch.cuda.Event per head GEMM launch, recorded on the launch
# stream immediately around the library call -- an interval taken around multi_level_heads() also holds this module's host-side
# preparation whenever the GPU is waiting for the host at that point.  None: nothing is recorded.
launch_events = None


def to_nhwc(x):
    """[B,C,H,W] tensor whose memory is NHWC (zero-copy when the producer already runs channels_last)."""
    x = x.float()
    if x.dim() != 4:
        raise ValueError('source maps must be [Batch, Channels, Height, Width]')
    return x.contiguous(memory_format=torch.channels_last)


def weight_khwc(w):
    """[N,Cin,3,3] parameter whose memory is [N,3,3,Cin]."""
    return w.float().contiguous(memory_format=torch.channels_last)


def _dp(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class _HeadsFn(torch.autograd.Function):
    """apply(x_0, ws_0, bs_0, wl_0, bl_0, x_1, ...) -> (scores [B, sum HW*nb*C], locs [B, sum HW*nb*4])"""

    @staticmethod
    def _level_array(levels, grads=None):
        arr = (_lib.HeadLevel * len(levels))()
        for i, lv in enumerate(levels):
            a = arr[i]
            a.x, a.h, a.w, a.cin = _dp(lv['x']), lv['H'], lv['W'], lv['cin']
            a.w_score, a.b_score, a.n_score = _dp(lv['ws']), _dp(lv['bs']), lv['ns']
            a.w_loc, a.b_loc, a.n_loc = _dp(lv['wl']), _dp(lv['bl']), lv['nl']
            a.scores_offset, a.locs_offset = lv['s_off'], lv['l_off']
            if grads is not None:
                gr = grads[i]
                a.dx, a.dw_score, a.db_score, a.dw_loc, a.db_loc = (_dp(gr['dx']), _dp(gr['dws']), _dp(gr['dbs']),
                                                                    _dp(gr['dwl']), _dp(gr['dbl']))
        return arr

    @staticmethod
    def forward(ctx, *args):
        lib = _lib.lib()
        L = len(args) // 5
        levels, saved, sinks = [], [], []
        s_off = l_off = 0
        for i in range(L):
            x, ws, bs, wl, bl = args[5 * i:5 * i + 5]
            _lib.require_cuda(x, ws, wl)
            sinks.append((ws, bs, wl, bl))   # the parameters themselves: their bucket slots are looked up in the backward
            x, ws, wl = to_nhwc(x), weight_khwc(ws), weight_khwc(wl)
            B, cin, H, W = x.shape
            if ws.shape[1:] != (cin, 3, 3) or wl.shape[1:] != (cin, 3, 3):
                raise ValueError(f'head {i}: weights {tuple(ws.shape)}/{tuple(wl.shape)} do not match Cin={cin}, 3x3')
            levels.append(dict(x=x, ws=ws, bs=None if bs is None else bs.float().contiguous(), wl=wl,
                               bl=None if bl is None else bl.float().contiguous(), B=B, cin=cin, H=H, W=W,
                               ns=ws.shape[0], nl=wl.shape[0], s_off=s_off, l_off=l_off))
            s_off += H * W * ws.shape[0]
            l_off += H * W * wl.shape[0]
        B = levels[0]['B']
        if any(lv['B'] != B for lv in levels):
            raise ValueError('all source maps must share the batch size')
        dev = levels[0]['x'].device
        scores = torch.empty((B, s_off), dtype=torch.float32, device=dev)
        locs = torch.empty((B, l_off), dtype=torch.float32, device=dev)
        arr = _HeadsFn._level_array(levels)
        events = launch_events
        if events is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if _lib.fast_mode == 'bf16x3':
            need = lib.ssdk_heads_fwd_fast_workspace_bytes(arr, L)
            ws = _lib.scratch(need, dev, 'heads_fwd_fast')
            _lib.check(lib.ssdk_heads_fwd_fast(arr, L, B, _dp(scores), s_off, _dp(locs), l_off, 3, _dp(ws), ws.numel(), _lib.current_stream()),
                       'ssdk_heads_fwd_fast')
        else:
            sk = _lib.scratch(lib.ssdk_heads_fwd_workspace_bytes(), dev, _lib.STREAMK_TAG, zeroed=True)   # (state kept by the library between calls)
            _lib.check(lib.ssdk_heads_fwd(arr, L, B, _dp(scores), s_off, _dp(locs), l_off, _dp(sk), sk.numel(), _lib.current_stream()), 'ssdk_heads_fwd')
        if events is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            events.append((e0, e1))
        # tensors go through save_for_backward (autograd's version counters then catch an in-place edit of a tapped source map
        # between forward and backward, as they do for torch's own conv); ctx keeps only shapes and offsets
        for lv in levels:
            saved += [lv.pop('x'), lv.pop('ws'), lv.pop('wl')]
            lv['has_bs'], lv['has_bl'] = lv.pop('bs') is not None, lv.pop('bl') is not None
        ctx.save_for_backward(*saved)
        ctx.levels = levels
        ctx.sinks = sinks
        ctx.totals = (s_off, l_off)
        return scores, locs

    @staticmethod
    def backward(ctx, dscores, dlocs):
        lib = _lib.lib()
        s_tot, l_tot = ctx.totals
        saved = ctx.saved_tensors
        levels = [dict(lv, x=saved[3 * i], ws=saved[3 * i + 1], wl=saved[3 * i + 2], bs=None, bl=None) for i, lv in enumerate(ctx.levels)]
        B = levels[0]['B']
        dev = levels[0]['x'].device
        dscores = (torch.zeros((B, s_tot), dtype=torch.float32, device=dev) if dscores is None else dscores.float().contiguous())
        dlocs = (torch.zeros((B, l_tot), dtype=torch.float32, device=dev) if dlocs is None else dlocs.float().contiguous())
        grads, out = [], []
        for i, lv in enumerate(levels):
            need_x, need_ws, need_bs, need_wl, need_bl = ctx.needs_input_grad[5 * i:5 * i + 5]
            x = lv['x']
            need_w = need_ws or need_wl
            need_b = (need_bs and lv['has_bs']) or (need_bl and lv['has_bl'])
            k_ws, k_bs, k_wl, k_bl = (grad_sink(t) for t in ctx.sinks[i])   # slots of a flat gradient bucket (distributed.GradBucket), when attached

            def dst(sink, like=None, n=None):
                if sink is not None:
                    return sink
                return torch.empty_like(like, memory_format=torch.channels_last) if like is not None else torch.empty((n,), dtype=torch.float32, device=dev)
            gr = dict(dx=torch.empty_like(x, memory_format=torch.channels_last) if need_x else None,
                      dws=dst(k_ws, like=lv['ws']) if need_w else None,
                      dwl=dst(k_wl, like=lv['wl']) if need_w else None,
                      dbs=dst(k_bs if lv['has_bs'] else None, n=lv['ns']) if need_b else None,
                      dbl=dst(k_bl if lv['has_bl'] else None, n=lv['nl']) if need_b else None)
            grads.append(gr)
            out += [gr['dx'], gr['dws'] if need_ws else None, gr['dbs'] if (need_bs and lv['has_bs']) else None,
                    gr['dwl'] if need_wl else None, gr['dbl'] if (need_bl and lv['has_bl']) else None]
        arr = _HeadsFn._level_array(levels, grads)
        need = lib.ssdk_heads_bwd_workspace_bytes(arr, len(levels), B)
        ws = _lib.scratch(need, dev, 'heads_bwd')
        _lib.check(lib.ssdk_heads_bwd(arr, len(levels), B, _dp(dscores), s_tot, _dp(dlocs), l_tot, _dp(ws), need,
                                      _lib.current_stream()), 'ssdk_heads_bwd')
        return tuple(out)


def multi_level_heads(sources_score, sources_loc, heads):
    """heads: nn.ModuleList of nn.ModuleDict({'score': Conv2d, 'loc': Conv2d}) (detector_builder.py:111-137).
    When the score and loc towers feed different maps (SharedConvPredictor) the two convs of a level run as two
    single-head GEMMs; otherwise they are fused along N."""
    args = []
    split = []
    for head, xs, xl in zip(heads, sources_score, sources_loc):
        if xs is xl:
            args += [xs, head['score'].weight, head['score'].bias, head['loc'].weight, head['loc'].bias]
            split.append(False)
        else:
            split.append(True)
    if not any(split):
        return _HeadsFn.apply(*args)
    # separate towers: run score heads and loc heads as two passes with an empty partner
    s_args, l_args = [], []
    for head, xs, xl in zip(heads, sources_score, sources_loc):
        empty_w = head['score'].weight.new_zeros((0,) + tuple(head['score'].weight.shape[1:]))
        empty_w_l = head['loc'].weight.new_zeros((0,) + tuple(head['loc'].weight.shape[1:]))
        s_args += [xs, head['score'].weight, head['score'].bias, empty_w, None]
        l_args += [xl, head['loc'].weight, head['loc'].bias, empty_w_l, None]
    scores, _ = _HeadsFn.apply(*s_args)
    locs, _ = _HeadsFn.apply(*l_args)
    return scores, locs
