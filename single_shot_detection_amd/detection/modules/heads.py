"""Multi-scale head convolutions on libssdk (csrc/conv.hip) -- the arithmetic of detection/detector.py:50-66.

One autograd function covers every pyramid level: it launches one implicit-GEMM per level that writes directly
into the concatenated ``scores [B, A*C]`` / ``locs [B, A*4]`` buffers (no permute / contiguous / cat), and its
backward produces the source-map gradients, the weight gradients and the bias gradients per level.
"""
import threading

import torch

from ... import _lib
from ...distributed import grad_sink


def set_fast_mode(mode):
    """Opt-in reduced-precision mode of the GEMMs, forward AND backward (the reference's analogue: apex AMP O1, bf/training/env.py:87-95,
    bf/training/callbacks.py:34-40).  ``None`` (default): exact fp32 on v_mfma_f32_32x32x2_f32.  ``'bf16x3'``: operands split into bf16
    pieces, three cross terms per product on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the head GEMM (ssdk_heads_fwd_fast),
    every ``ops.conv2d`` of at least ~1 GFLOP whose input has Cin % 32 == 0 (ssdk_conv2d_fwd_fast: RetinaNet's towers, necks, the large
    layers of a pyramid tail), and in the backward pass the dense stride-1 data gradients and every weight gradient (ssdk_heads_bwd_fast,
    ssdk_conv2d_bwd_fast); bias gradients, the sparse / strided data gradients, norms and losses stay fp32.  Outputs and gradients within
    ~1e-5 of their scale.  The heads need Cin % 32 == 0 on every level (the forward call raises otherwise); other convolutions stay fp32
    where it does not hold.  Returns the previous setting."""
    if mode not in (None, 'bf16x3'):
        raise ValueError(f'fast mode {mode!r}: None or "bf16x3"')
    prev, _lib.fast_mode = _lib.fast_mode, mode
    return prev


class fast_mode(object):
    """``with heads.fast_mode('bf16x3'):`` -- ``set_fast_mode`` for the calls inside the block; the previous mode comes back on exit."""

    def __init__(self, mode):
        if mode not in (None, 'bf16x3'):
            raise ValueError(f'fast mode {mode!r}: None or "bf16x3"')
        self.mode = mode

    def __enter__(self):
        self.prev = set_fast_mode(self.mode)
        return self

    def __exit__(self, *exc):
        set_fast_mode(self.prev)
        return False


# Measurement hook (bench.py): a list that receives one (start event, end event, ((H, W, Cin, N), ...) of the launch's levels) per head GEMM
# launch, recorded on the launch stream immediately around the library call -- an interval taken around multi_level_heads() also holds this module's host-side
# preparation whenever the GPU is waiting for the host at that point.  None: nothing is recorded.
launch_events = None


# What the producer of the heads' upstream gradient knows about its sparsity (MultiboxLoss's backward under hard-negative mining: ~4 % of
# the anchors carry a gradient): set by the producer inside a backward pass, taken -- once -- by the heads' backward of the SAME pass,
# and only for the very tensors the producer returned (same storage, same version: anything autograd added on the way is another tensor).
class RowHint(object):
    __slots__ = ('task', 'key', 'mask')

    def __init__(self, dscores, dlocs, mask):
        self.task = torch._C._current_graph_task_id()
        self.key = (dscores.data_ptr(), dscores._version, tuple(dscores.shape), dlocs.data_ptr(), dlocs._version, tuple(dlocs.shape))
        self.mask = mask   # uint8 [B, A]: 0 = the anchor's dscores and dlocs rows are zeros


_hint_slot = threading.local()   # per autograd thread (torch runs one per device): threads cannot consume one another's hint
row_hints_taken = 0   # (tests / diagnostics: heads backward calls that ran with a producer's row mask)


def set_row_hint(dscores, dlocs, mask):
    """Called by the producer of the heads' upstream gradient inside its backward (MultiboxLoss): ``mask`` uint8 [B, A], 0 = the anchor's
    rows of ``dscores`` / ``dlocs`` are zeros."""
    _hint_slot.hint = RowHint(dscores, dlocs, mask)


def take_row_hint(dscores, dlocs):
    """The row mask for exactly these gradient tensors in this backward pass, or None; the hint is consumed either way."""
    h, _hint_slot.hint = getattr(_hint_slot, 'hint', None), None
    if h is None or dscores is None or dlocs is None or h.task != torch._C._current_graph_task_id():
        return None
    if h.key != (dscores.data_ptr(), dscores._version, tuple(dscores.shape), dlocs.data_ptr(), dlocs._version, tuple(dlocs.shape)):
        return None
    if h.mask.shape[0] != dscores.shape[0] or dlocs.shape[1] != 4 * h.mask.shape[1]:
        return None
    global row_hints_taken
    row_hints_taken += 1
    return h.mask


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


def _level_array(levels, grads=None):
    arr = (_lib.HeadLevel * len(levels))()
    for i, lv in enumerate(levels):
        a = arr[i]
        a.x, a.h, a.w, a.cin = _dp(lv['x']), lv['H'], lv['W'], lv['cin']
        a.w_score, a.b_score, a.n_score = _dp(lv['ws']), _dp(lv['bs']), lv['ns']
        a.w_loc, a.b_loc, a.n_loc = _dp(lv['wl']), _dp(lv['bl']), lv['nl']
        # (a single head -- no loc part -- carries its anchor-type count where the locs offset would be: include/ssdk.h ssdk_head_level)
        a.scores_offset, a.locs_offset = lv['s_off'], (lv['l_off'] if lv['nl'] else int(lv.get('nb_hint') or 0))
        if grads is not None:
            gr = grads[i]
            a.dx, a.dw_score, a.db_score, a.dw_loc, a.db_loc = (_dp(gr['dx']), _dp(gr['dws']), _dp(gr['dbs']),
                                                                _dp(gr['dwl']), _dp(gr['dbl']))
    return arr


def _parse_levels(args, s_off=0, l_off=0, nb_hints=None):
    """(x, w_score, b_score, w_loc, b_loc) * L -> level dicts (NHWC maps, [N,3,3,Cin] weights, offsets into one image's output row),
    the parameters themselves (their gradient-bucket slots are looked up in the backward), and the offsets behind the last level."""
    levels, sinks = [], []
    for i in range(len(args) // 5):
        x, ws, bs, wl, bl = args[5 * i:5 * i + 5]
        _lib.require_cuda(x, ws, wl)
        sinks.append((ws, bs, wl, bl))
        x, ws, wl = to_nhwc(x), weight_khwc(ws), weight_khwc(wl)
        B, cin, H, W = x.shape
        if ws.shape[1:] != (cin, 3, 3) or wl.shape[1:] != (cin, 3, 3):
            raise ValueError(f'head {i}: weights {tuple(ws.shape)}/{tuple(wl.shape)} do not match Cin={cin}, 3x3')
        levels.append(dict(x=x, ws=ws, bs=None if bs is None else bs.float().contiguous(), wl=wl,
                           bl=None if bl is None else bl.float().contiguous(), B=B, cin=cin, H=H, W=W,
                           ns=ws.shape[0], nl=wl.shape[0], s_off=s_off, l_off=l_off,
                           nb_hint=(nb_hints[i] if nb_hints is not None else 0)))
        s_off += H * W * ws.shape[0]
        l_off += H * W * wl.shape[0]
    if any(lv['B'] != levels[0]['B'] for lv in levels):
        raise ValueError('all source maps must share the batch size')
    return levels, sinks, s_off, l_off


def _launch_forward(levels, scores, locs, max_workgroups=0):
    """The grouped forward GEMM of ``levels`` into their slices of the rows of ``scores`` / ``locs`` on the current stream."""
    lib = _lib.lib()
    L, B, dev = len(levels), levels[0]['B'], levels[0]['x'].device
    s_tot, l_tot = scores.shape[1], locs.shape[1]
    arr = _level_array(levels)
    events = launch_events
    if events is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if _lib.fast_mode == 'bf16x3':
        need = lib.ssdk_heads_fwd_fast_workspace_bytes(arr, L)
        ws = _lib.scratch(need, dev, 'heads_fwd_fast')
        _lib.check(lib.ssdk_heads_fwd_fast(arr, L, B, _dp(scores), s_tot, _dp(locs), l_tot, 3, _dp(ws), ws.numel(), _lib.current_stream()),
                   'ssdk_heads_fwd_fast')
    else:
        sk = _lib.scratch(lib.ssdk_heads_fwd_workspace_bytes(), dev, _lib.STREAMK_TAG, zeroed=True)   # (state kept by the library between calls)
        _lib.check(lib.ssdk_heads_fwd_ex(arr, L, B, _dp(scores), s_tot, _dp(locs), l_tot, int(max_workgroups), _dp(sk), sk.numel(),
                                         _lib.current_stream()), 'ssdk_heads_fwd')
    if events is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        events.append((e0, e1, tuple((lv['H'], lv['W'], lv['cin'], lv['ns'] + lv['nl']) for lv in levels)))


def _stash_levels(ctx, levels, sinks):
    # tensors go through save_for_backward (autograd's version counters then catch an in-place edit of a tapped source map
    # between forward and backward, as they do for torch's own conv); ctx keeps only shapes and offsets
    saved = []
    for lv in levels:
        saved += [lv.pop('x'), lv.pop('ws'), lv.pop('wl')]
        lv['has_bs'], lv['has_bl'] = lv.pop('bs') is not None, lv.pop('bl') is not None
    ctx.save_for_backward(*saved)
    ctx.levels = levels
    ctx.sinks = sinks


def _launch_backward(ctx, dscores, dlocs, needs, s_tot, l_tot, row_mask=None):
    """dgrad + wgrad + dbias of ctx's levels from their slices of the rows of dscores / dlocs; returns the flat gradient tuple
    (dx, dw_score, db_score, dw_loc, db_loc) * L.  ``needs``: ctx.needs_input_grad of those 5 L arguments.  ``row_mask``: uint8 [B, A],
    0 = the producer of the gradient guarantees that anchor's rows to be zeros (take_row_hint)."""
    lib = _lib.lib()
    saved = ctx.saved_tensors
    levels = [dict(lv, x=saved[3 * i], ws=saved[3 * i + 1], wl=saved[3 * i + 2], bs=None, bl=None) for i, lv in enumerate(ctx.levels)]
    B = levels[0]['B']
    dev = levels[0]['x'].device
    grads, out = [], []
    for i, lv in enumerate(levels):
        need_x, need_ws, need_bs, need_wl, need_bl = needs[5 * i:5 * i + 5]
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
    arr = _level_array(levels, grads)
    # fast mode: the dense data gradients on the split-bf16 GEMM; sparse forms, weight / bias gradients unchanged
    fast = _lib.fast_mode == 'bf16x3'
    need = (lib.ssdk_heads_bwd_fast_workspace_bytes if fast else lib.ssdk_heads_bwd_workspace_bytes)(arr, len(levels), B)
    ws = _lib.scratch(need, dev, 'heads_bwd')
    _lib.check(lib.ssdk_heads_bwd_ex(arr, len(levels), B, _dp(dscores), s_tot, _dp(dlocs), l_tot, _dp(row_mask),
                                     0 if row_mask is None else row_mask.shape[1], 3 if fast else 0, _dp(ws), need, _lib.current_stream()),
               'ssdk_heads_bwd')
    return out


def peek_row_hint_one(d):
    """Single-head calls (SharedConvPredictor: the score heads and the loc heads run as two calls): the producer's row mask when ``d`` is
    either of the two gradient tensors it was set for.  Not consumed: the partner call wants it too; the next backward pass replaces it."""
    h = getattr(_hint_slot, 'hint', None)
    if h is None or d is None or h.task != torch._C._current_graph_task_id():
        return None
    k = (d.data_ptr(), d._version, tuple(d.shape))
    if k != h.key[:3] and k != h.key[3:]:
        return None
    if h.mask.shape[0] != d.shape[0]:
        return None
    global row_hints_taken
    row_hints_taken += 1
    return h.mask


class _HeadsFn(torch.autograd.Function):
    """apply(nb_hints, x_0, ws_0, bs_0, wl_0, bl_0, x_1, ...) -> (scores [B, sum HW*nb*C], locs [B, sum HW*nb*4]); nb_hints: None, or
    for single-head levels (empty loc weights) the anchor types per pixel, which such a level cannot tell from its own shapes"""

    @staticmethod
    def forward(ctx, nb_hints, *args):
        levels, sinks, s_off, l_off = _parse_levels(args, nb_hints=nb_hints)
        B, dev = levels[0]['B'], levels[0]['x'].device
        scores = torch.empty((B, s_off), dtype=torch.float32, device=dev)
        locs = torch.empty((B, l_off), dtype=torch.float32, device=dev)
        _launch_forward(levels, scores, locs)
        _stash_levels(ctx, levels, sinks)
        ctx.totals = (s_off, l_off)
        return scores, locs

    @staticmethod
    def backward(ctx, dscores, dlocs):
        s_tot, l_tot = ctx.totals
        B, dev = ctx.levels[0]['B'], ctx.saved_tensors[0].device
        single = l_tot == 0 and all(lv.get('nb_hint') for lv in ctx.levels)
        mask = peek_row_hint_one(dscores) if single else take_row_hint(dscores, dlocs)
        dscores = (torch.zeros((B, s_tot), dtype=torch.float32, device=dev) if dscores is None else dscores.float().contiguous())
        dlocs = (torch.zeros((B, l_tot), dtype=torch.float32, device=dev) if dlocs is None else dlocs.float().contiguous())
        return (None,) + tuple(_launch_backward(ctx, dscores, dlocs, ctx.needs_input_grad[1:], s_tot, l_tot, mask))


# ---- the same heads as TWO independent autograd nodes (dependency split) -------------------------------------------------------------
# SSD: the first pyramid levels are backbone taps, the rest come out of the pyramid tail (extras), a chain of small convolutions that
# cannot fill the chip.  Levels 0-1 carry ~93 % of the head FLOPs and do not depend on the tail, so their GEMM (and, in the backward
# pass, their data / weight gradients) can run BESIDE the tail on another stream.  Both parts write their slices of the same
# [B, A*C] / [B, A*4] rows; a join node hands the buffers to the loss and, in the backward pass, hands the loss' gradient rows to
# both parts (no copy: each part's pack kernel reads its own slices).
class _HeadsShared(object):
    """What the parts of one split heads call share: the output rows while the forward pass runs (dropped once the join has handed them
    on: the join node's outputs would otherwise keep the autograd graph alive through this object), and in the backward pass the
    gradient rows the join received."""
    __slots__ = ('scores', 'locs', 's_tot', 'l_tot', 'dscores', 'dlocs', 'join_event', 'join_stream', 'pending', 'first_done', 'order', 'row_mask', 'parts_left')

    def __init__(self, scores, locs):
        self.scores, self.locs = scores, locs
        self.s_tot, self.l_tot = scores.shape[1], locs.shape[1]
        self.dscores = self.dlocs = self.join_event = self.join_stream = None
        self.pending = []        # levels of parts that left their GEMM to the join (ONE grouped launch of all levels)
        self.first_done = None   # backward: event behind the part that ran first (order = True: the other part waits for it)
        self.order = False
        self.row_mask = None     # backward: the gradient producer's row mask (take_row_hint), for every part
        self.parts_left = 0      # backward: parts that have not run yet in this pass (the last one drops the gradient rows)


class _HeadsPartFn(torch.autograd.Function):
    """apply(shared, s_off, l_off, max_workgroups, x_i, ws_i, bs_i, wl_i, bl_i, ...) -> token (an empty tensor that carries the
    dependency to _HeadsJoinFn): the levels' GEMM into shared.scores / shared.locs from column s_off / l_off on, on the CURRENT stream."""

    @staticmethod
    def forward(ctx, shared, s_off, l_off, max_workgroups, *args):
        levels, sinks, _, _ = _parse_levels(args, s_off, l_off)
        if levels[0]['B'] != shared.scores.shape[0]:
            raise ValueError('all source maps must share the batch size')
        if max_workgroups < 0:   # the join launches these levels together with the other parts'
            shared.pending += [dict(lv) for lv in levels]
        else:
            _launch_forward(levels, shared.scores, shared.locs, max_workgroups)
        _stash_levels(ctx, levels, sinks)
        ctx.shared = shared
        return shared.scores.new_empty((0,))

    @staticmethod
    def backward(ctx, _token_grad):
        sh = ctx.shared
        if sh.dscores is None:
            raise RuntimeError('split heads: the join node has not delivered the gradient rows (backward through a part alone)')
        cur = torch.cuda.current_stream()
        if cur != sh.join_stream:
            # this part runs on another stream than the loss' backward: wait for the rows, and keep them alive for this stream's reads
            cur.wait_event(sh.join_event)
            sh.dscores.record_stream(cur)
            sh.dlocs.record_stream(cur)
        if sh.order and sh.first_done is not None:
            cur.wait_event(sh.first_done)   # (the part in front of the pyramid tail's backward chain had the chip to itself)
        if sh.row_mask is not None and cur != sh.join_stream:
            sh.row_mask.record_stream(cur)
        out = _launch_backward(ctx, sh.dscores, sh.dlocs, ctx.needs_input_grad[4:], sh.s_tot, sh.l_tot, sh.row_mask)
        if sh.order and sh.first_done is None:
            sh.first_done = torch.cuda.Event()
            sh.first_done.record(cur)
        sh.parts_left -= 1
        if sh.parts_left <= 0:   # the gradient rows (88 MB at SSD-300, batch 32) are not held until the graph is freed
            sh.dscores = sh.dlocs = sh.row_mask = None
        return (None, None, None, None) + tuple(out)


class _HeadsJoinFn(torch.autograd.Function):
    """apply(shared, token_0, token_1, ...) -> (scores, locs): all parts have been enqueued (the caller has joined their streams)."""

    @staticmethod
    def forward(ctx, shared, *tokens):
        ctx.shared = shared
        ctx.n = len(tokens)
        ctx.meta = (tuple(shared.scores.shape), tuple(shared.locs.shape), shared.scores.device)
        scores, locs = shared.scores, shared.locs
        if shared.pending:
            _launch_forward(sorted(shared.pending, key=lambda lv: lv['s_off']), scores, locs)
            shared.pending = []
        shared.scores = shared.locs = None
        return scores, locs

    @staticmethod
    def backward(ctx, dscores, dlocs):
        sh = ctx.shared
        s_shape, l_shape, dev = ctx.meta
        sh.row_mask = take_row_hint(dscores, dlocs)
        sh.first_done = None      # (a second backward pass through a retained graph orders its parts on its OWN event)
        sh.parts_left = ctx.n
        sh.dscores = torch.zeros(s_shape, dtype=torch.float32, device=dev) if dscores is None else dscores.float().contiguous()
        sh.dlocs = torch.zeros(l_shape, dtype=torch.float32, device=dev) if dlocs is None else dlocs.float().contiguous()
        sh.join_stream = torch.cuda.current_stream()
        sh.join_event = torch.cuda.Event()
        sh.join_event.record(sh.join_stream)
        return (None,) + tuple(sh.dscores.new_empty((0,)) for _ in range(ctx.n))


def multi_level_heads_split(sources, heads, split, run_tail, side_stream=None, main_workgroups=0, side_workgroups=0, one_launch=False,
                            ordered_backward=False):
    """Heads over ``sources[:split]`` (backbone taps) and over the levels ``run_tail()`` returns (the pyramid tail's outputs), as two
    autograd nodes; with a ``side_stream`` the tail and its levels' heads run there -- in the backward pass too (autograd runs a node on
    its forward stream): the tail's chain of small kernels is differentiated beside the first levels' data / weight gradients.
    ``one_launch``: the forward GEMM of ALL levels is one grouped launch on the current stream behind the tail (what multi_level_heads
    does; only the backward pass is split).  Otherwise the two parts launch their own GEMMs, the first levels' beside the tail
    (``main_workgroups``: persistent workgroups of that GEMM -- fewer than the chip's 512 slots leave room for the tail's kernels; 0: the
    library's default).  ``ordered_backward``: the first levels' backward waits for the tail levels' (which unblocks the tail's chain).
    Returns (scores, locs, all sources)."""
    sources = list(sources)
    first = sources[:split]
    main = torch.cuda.current_stream()
    forked = side_stream is not None and side_stream != main

    def part_args(xs, hs):
        a = []
        for x, h in zip(xs, hs):
            a += [x, h['score'].weight, h['score'].bias, h['loc'].weight, h['loc'].bias]
        return a

    if forked:
        side_stream.wait_stream(main)
        with torch.cuda.stream(side_stream):
            rest = list(run_tail())
    else:
        rest = list(run_tail())
    allsrc = first + rest
    assert len(allsrc) == len(heads), (len(allsrc), len(heads))
    s_offs, l_offs, s_off, l_off = [], [], 0, 0
    for x, h in zip(allsrc, heads):
        s_offs.append(s_off)
        l_offs.append(l_off)
        s_off += x.shape[2] * x.shape[3] * h['score'].weight.shape[0]
        l_off += x.shape[2] * x.shape[3] * h['loc'].weight.shape[0]
    B, dev = allsrc[0].shape[0], allsrc[0].device
    shared = _HeadsShared(torch.empty((B, s_off), dtype=torch.float32, device=dev), torch.empty((B, l_off), dtype=torch.float32, device=dev))
    shared.order = bool(ordered_backward)
    # (the rows were allocated after the fork, but nothing has been enqueued on the current stream since: whatever used their memory
    # before was enqueued in front of the fork point the side stream waited for)
    # order of the applies = reverse order of the backward calls: the tail levels' part is differentiated first, then the first levels',
    # then the tail's chain (its nodes are older than both)
    tok_a = _HeadsPartFn.apply(shared, 0, 0, -1 if one_launch else main_workgroups, *part_args(first, heads[:split]))
    if forked:
        with torch.cuda.stream(side_stream):
            tok_b = _HeadsPartFn.apply(shared, s_offs[split], l_offs[split], -1 if one_launch else side_workgroups, *part_args(rest, heads[split:]))
        main.wait_stream(side_stream)
        shared.scores.record_stream(side_stream)   # (allocated on this stream, written on the other)
        shared.locs.record_stream(side_stream)
    else:
        tok_b = _HeadsPartFn.apply(shared, s_offs[split], l_offs[split], -1 if one_launch else side_workgroups, *part_args(rest, heads[split:]))
    scores, locs = _HeadsJoinFn.apply(shared, tok_a, tok_b)
    return scores, locs, allsrc


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
        return _HeadsFn.apply(None, *args)
    # separate towers: run score heads and loc heads as two passes with an empty partner
    s_args, l_args = [], []
    for head, xs, xl in zip(heads, sources_score, sources_loc):
        empty_w = head['score'].weight.new_zeros((0,) + tuple(head['score'].weight.shape[1:]))
        empty_w_l = head['loc'].weight.new_zeros((0,) + tuple(head['loc'].weight.shape[1:]))
        s_args += [xs, head['score'].weight, head['score'].bias, empty_w, None]
        l_args += [xl, head['loc'].weight, head['loc'].bias, empty_w_l, None]
    # (anchor types per pixel: what the partner head would have told the library -- the ordered sparse backward needs it)
    hints = tuple(head['loc'].weight.shape[0] // 4 for head in heads)
    scores, _ = _HeadsFn.apply(hints, *s_args)
    locs, _ = _HeadsFn.apply(hints, *l_args)
    return scores, locs
