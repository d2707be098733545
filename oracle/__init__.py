"""ctypes binding of the CPU oracle (``oracle/ssdk_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg, never by the product package ``single_shot_detection_amd``.
All arrays are numpy, C-contiguous; float32 unless stated.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libssdk_oracle.so')


def build(force=False):
    src = os.path.join(_HERE, 'ssdk_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libssdk_oracle.so'], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_nms_hard.restype = C.c_int
        _lib.orc_nms_soft.restype = C.c_int
        _lib.orc_anchors_ssd_level.restype = C.c_int
        _lib.orc_anchors_retina_level.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def max_threads():
    return lib().orc_max_threads()


def set_threads(n):
    """OpenMP threads of the oracle's parallel loops from now on (bench.py's cpu_baseline probes {16, 32, 64, 128})."""
    lib().orc_set_threads(C.c_int(int(n)))


def pack_gt(gt_list, stride=None):
    """list[B] of [G_i, >=6] -> (rows [sum G, stride] f32, offsets int32 [B+1])."""
    stride = stride or max([6] + [g.shape[1] for g in gt_list if g.size])
    rows = [np.asarray(g, dtype=np.float32).reshape(-1, g.shape[1] if g.size else stride)[:, :stride] for g in gt_list]
    off = np.zeros(len(gt_list) + 1, dtype=np.int32)
    off[1:] = np.cumsum([r.shape[0] for r in rows])
    flat = np.concatenate(rows, axis=0) if off[-1] else np.zeros((0, stride), np.float32)
    return _f32(flat), off


def to_corners(box):
    box = _f32(box)
    out = np.empty_like(box)
    lib().orc_to_corners(_p(box), _p(out), C.c_int64(box.size // 4))
    return out


def iou(gt, corner_anchors):
    gt = _f32(gt)
    corner = _f32(corner_anchors)
    out = np.empty((gt.shape[0], corner.shape[0]), np.float32)
    lib().orc_iou(_p(gt), C.c_int(gt.shape[0]), C.c_int(gt.shape[1]), _p(corner), C.c_int64(corner.shape[0]), _p(out))
    return out


def match_per_prediction(weights, matched, unmatched=None, force=True):
    w = _f32(weights)
    unmatched = matched if unmatched is None else unmatched
    idx = np.empty(w.shape[1], np.int64)
    lib().orc_match_per_prediction(_p(w), C.c_int(w.shape[0]), C.c_int64(w.shape[1]), C.c_float(matched),
                                   C.c_float(unmatched), C.c_int(int(force)), _p(idx))
    return idx


def encode_ground_truth(gt_list, anchors, matched, unmatched, return_box_idx=False):
    rows, off = pack_gt(gt_list)
    anchors = _f32(anchors)
    B, A = len(gt_list), anchors.shape[0]
    target = np.empty((B, A, 6), np.float32)
    bidx = np.empty((B, A), np.int32) if return_box_idx else None
    lib().orc_encode_ground_truth(_p(rows), _p(off), C.c_int(B), C.c_int(rows.shape[1]), _p(anchors), C.c_int64(A),
                                  C.c_float(matched), C.c_float(unmatched), _p(target), _p(bidx))
    return (target, bidx) if return_box_idx else target


def to_centroids_inplace(box):
    assert box.dtype == np.float32 and box.flags.c_contiguous
    lib().orc_to_centroids_inplace(_p(box), C.c_int64(box.size // 4))
    return box


def encode_box_inplace(boxes, priors, xy_scale=10.0, wh_scale=5.0, eps=1e-8):
    assert boxes.dtype == np.float32 and boxes.flags.c_contiguous
    priors = _f32(priors)
    B = boxes.shape[0] if boxes.ndim == 3 else 1
    lib().orc_encode_box_inplace(_p(boxes), _p(priors), C.c_int(B), C.c_int64(priors.shape[0]), C.c_float(xy_scale),
                                 C.c_float(wh_scale), C.c_float(eps))
    return boxes


def encode_box(boxes, priors, xy_scale=10.0, wh_scale=5.0, eps=1e-8):
    boxes, priors = _f32(boxes), _f32(priors)
    out = np.empty_like(boxes)
    B = boxes.shape[0] if boxes.ndim == 3 else 1
    lib().orc_encode_box(_p(boxes), _p(priors), _p(out), C.c_int(B), C.c_int64(priors.shape[0]), C.c_float(xy_scale),
                         C.c_float(wh_scale), C.c_float(eps))
    return out


def decode_box(locs, priors, xy_scale=10.0, wh_scale=5.0):
    locs, priors = _f32(locs), _f32(priors)
    out = np.empty_like(locs)
    B = locs.shape[0] if locs.ndim == 3 else 1
    lib().orc_decode_box(_p(locs), _p(priors), _p(out), C.c_int(B), C.c_int64(priors.shape[0]), C.c_float(xy_scale),
                         C.c_float(wh_scale))
    return out


def hard_negative_mining(scores, target, ratio, min_neg, return_bgloss=False):
    target = _f32(target)
    B, A = target.shape[:2]
    scores = _f32(scores).reshape(B, A, -1)
    mask = np.empty((B, A), np.uint8)
    bg = np.empty((B, A), np.float32) if return_bgloss else None
    lib().orc_hard_negative_mining(_p(scores), _p(target), C.c_int(B), C.c_int64(A), C.c_int(scores.shape[2]),
                                   C.c_double(ratio), C.c_int64(min_neg), _p(mask), _p(bg))
    mask = mask.astype(bool)
    return (mask, bg) if return_bgloss else mask


def naive_sampler(scores, target):
    cls = np.asarray(target)[..., 4].astype(np.int64)
    return (cls != 0) & (cls != -1)


def multibox_loss(scores, locs, anchors, target, sampled, kind='ce', gamma=2.0, alpha=0.25, reduce_mean=True,
                  cls_w=1.0, loc_w=1.0, xy_scale=10.0, wh_scale=5.0, eps=1e-8, beta=1.0, grads=True):
    """Returns (values[3] f64 = loss, class_loss, loc_loss; dscores [B,A,C]; dlocs [B,A,4]).  MUTATES target."""
    assert target.dtype == np.float32 and target.flags.c_contiguous
    B, A = target.shape[:2]
    scores = _f32(scores).reshape(B, A, -1)
    locs = _f32(locs).reshape(B, A, 4)
    anchors = _f32(anchors)
    Cn = scores.shape[2]
    smp = np.ascontiguousarray(sampled, dtype=np.uint8)
    out3 = np.zeros(3, np.float64)
    ds = np.empty_like(scores) if grads else None
    dl = np.empty_like(locs) if grads else None
    if kind == 'ce':
        lib().orc_multibox_loss_ce(_p(scores), _p(locs), _p(anchors), _p(target), _p(smp), C.c_int(B), C.c_int64(A),
                                   C.c_int(Cn), C.c_float(cls_w), C.c_float(loc_w), C.c_float(xy_scale),
                                   C.c_float(wh_scale), C.c_float(eps), C.c_float(beta), _p(out3), _p(ds), _p(dl))
    elif kind == 'focal':
        lib().orc_multibox_loss_focal(_p(scores), _p(locs), _p(anchors), _p(target), _p(smp), C.c_int(B), C.c_int64(A),
                                      C.c_int(Cn), C.c_float(gamma), C.c_float(alpha), C.c_int(int(reduce_mean)),
                                      C.c_float(cls_w), C.c_float(loc_w), C.c_float(xy_scale), C.c_float(wh_scale),
                                      C.c_float(eps), C.c_float(beta), _p(out3), _p(ds), _p(dl))
    else:
        raise ValueError(kind)
    return out3, ds, dl


CLS_KINDS = {'ce': 0, 'focal': 1, 'softmax_focal': 2, 'ce_soft': 3, 'bce_soft': 4}
LOC_KINDS = {'smooth_l1': 0, 'giou': 1}


def multibox_loss_ex(scores, locs, anchors, target, sampled, cls_kind='ce', loc_kind='smooth_l1', gamma=2.0, alpha=-1.0, epsilon=0.0,
                     reduce_mean=True, cls_w=1.0, loc_w=1.0, xy_scale=10.0, wh_scale=5.0, eps=1e-8, beta=1.0, grads=True):
    """The remaining selectable losses (SoftmaxFocal, soft-target CE / BCE, GIoU).  alpha < 0 means None.  MUTATES target
    unless loc_kind == 'giou'."""
    assert target.dtype == np.float32 and target.flags.c_contiguous
    B, A = target.shape[:2]
    scores = _f32(scores).reshape(B, A, -1)
    locs = _f32(locs).reshape(B, A, 4)
    anchors = _f32(anchors)
    smp = np.ascontiguousarray(sampled, dtype=np.uint8)
    out3 = np.zeros(3, np.float64)
    ds = np.empty_like(scores) if grads else None
    dl = np.empty_like(locs) if grads else None
    lib().orc_multibox_loss_ex(_p(scores), _p(locs), _p(anchors), _p(target), _p(smp), C.c_int(B), C.c_int64(A), C.c_int(scores.shape[2]),
                               C.c_int(CLS_KINDS[cls_kind]), C.c_int(LOC_KINDS[loc_kind]), C.c_float(gamma), C.c_float(alpha),
                               C.c_float(epsilon), C.c_int(int(reduce_mean)), C.c_float(cls_w), C.c_float(loc_w), C.c_float(xy_scale),
                               C.c_float(wh_scale), C.c_float(eps), C.c_float(beta), _p(out3), _p(ds), _p(dl))
    return out3, ds, dl


def nms_hard(boxes, scores, thr):
    boxes, scores = _f32(boxes), _f32(scores)
    picked = np.empty(max(1, scores.shape[0]), np.int32)
    n = lib().orc_nms_hard(_p(boxes), _p(scores), C.c_int(scores.shape[0]), C.c_float(thr), _p(picked))
    return picked[:n].astype(np.int64)


def nms_soft(boxes, scores, score_thr, sigma=0.5):
    boxes, scores = _f32(boxes), _f32(scores)
    picked = np.empty(max(1, scores.shape[0]), np.int32)
    n = lib().orc_nms_soft(_p(boxes), _p(scores), C.c_int(scores.shape[0]), C.c_float(score_thr), C.c_float(sigma), _p(picked))
    return picked[:n].astype(np.int64)


def postprocess(scores, locs, priors, softmax=True, score_thr=0.01, max_per_class=100, nms_thr=0.45, soft=False,
                sigma=0.5, max_total=200, xy_scale=10.0, wh_scale=5.0, return_cand=False):
    priors = _f32(priors)
    A = priors.shape[0]
    locs = _f32(locs)
    B = locs.shape[0]
    scores = _f32(scores).reshape(B, A, -1)
    Cn = scores.shape[2]
    ncls = Cn - 1 if softmax else Cn
    mpc = max_per_class if max_per_class else 0
    mt = max_total if max_total else 0
    cap = mt if mt else ncls * (mpc if mpc else A)
    out = np.zeros((B, cap, 6), np.float32)
    counts = np.zeros(B, np.int32)
    cand = np.zeros(B, np.int64)
    lib().orc_postprocess(_p(scores), _p(locs.reshape(B, A, 4)), _p(priors), C.c_int(B), C.c_int64(A), C.c_int(Cn),
                          C.c_int(int(softmax)), C.c_float(score_thr), C.c_int(mpc), C.c_float(nms_thr),
                          C.c_int(int(soft)), C.c_float(sigma), C.c_int(mt), C.c_float(xy_scale), C.c_float(wh_scale),
                          _p(out), C.c_int(cap), _p(counts), _p(cand))
    res = [out[i, :counts[i]].copy() for i in range(B)]
    return (res, cand) if return_cand else res


def linspace_f32(start, end, steps, use_fma=True):
    out = np.empty(steps, np.float32)
    lib().orc_linspace_f32(C.c_float(start), C.c_float(end), C.c_int64(steps), C.c_int(int(use_fma)), _p(out))
    return out


def anchors(cfg_anchor, size, levels, use_fma=True):
    """Anchors [A,4] for an ``anchor_generator`` dict of a reference sample file and levels [(cin, h, nb)]."""
    p = dict(cfg_anchor)
    outs = []
    if p['type'] == 'ssd':
        L = p['num_scales']
        scales = linspace_f32(p['min_scale'], p['max_scale'], L + 1, use_fma)  # ssd.py:33
        for i, (_, h, nb) in enumerate(levels):
            r = np.asarray(p['aspect_ratios'][i], dtype=np.float64)
            out = np.empty((h, h, nb, 4), np.float32)
            n = lib().orc_anchors_ssd_level(_p(r), C.c_int(len(r)), C.c_float(scales[i]), C.c_float(scales[i + 1]),
                                            C.c_int(size), C.c_int(size), C.c_int(h), C.c_int(h), C.c_int(int(use_fma)), _p(out))
            assert n == nb, (n, nb)
            outs.append(out.reshape(-1, 4))
    elif p['type'] == 'retina_net':
        r = np.asarray(p['aspect_ratios'], dtype=np.float64)
        for lvl, (_, h, nb) in zip(range(p['min_level'], p['max_level'] + 1), levels):
            out = np.empty((h, h, nb, 4), np.float32)
            n = lib().orc_anchors_retina_level(_p(r), C.c_int(len(r)), C.c_int(lvl), C.c_double(p['scale']),
                                               C.c_int(p['scales_per_level']), C.c_int(size), C.c_int(size), C.c_int(h),
                                               C.c_int(h), C.c_int(int(use_fma)), _p(out))
            assert n == nb, (n, nb)
            outs.append(out.reshape(-1, 4))
    else:
        raise ValueError(p['type'])
    return np.concatenate(outs, axis=0)


def ssd_anchor_generator(img_wh, fmap_wh, aspect_ratios, min_scale=None, max_scale=None, min_size=None, max_size=None, step=None,
                         offset=(.5, .5), num_branches=1, flip=True, use_fma=True):
    """detection/anchor_generators/ssd.py:55-151 SsdAnchorGenerator(...).generate for one level, every constructor mode:
    [H, W, num_boxes, 4] float32.  Arithmetic as the reference mixes it: fp32 linspace tables (ssd.py:102-104), fp32 products with
    python scalars rounded to fp32 first (:125, :132-133), double square root of an fp32 product (:135-136), fp32 linspace of the
    centres from python-float ends (:138-139)."""
    import math
    img_w, img_h = img_wh
    W, H = fmap_wh
    ars = []
    for ar in aspect_ratios:                       # :86-92
        ars.append(ar)
        if ar > 1.0 and flip:
            ars.append(1.0 / ar)
    nr = len(ars) + 1
    if min_size is not None and max_size is not None:
        lin = linspace_f32(float(min_size), float(max_size), num_branches + 1, use_fma)
        sizes = np.stack([lin, lin], 1)
    else:
        lin = linspace_f32(float(np.float32(min_scale)), float(np.float32(max_scale)), num_branches + 1, use_fma)
        sizes = np.stack([lin * np.float32(img_w), lin * np.float32(img_h)], 1)
    hws = np.empty((nr * num_branches, 2), np.float32)
    for j in range(num_branches):
        mn, mx = sizes[j], sizes[j + 1]
        for i, r in enumerate(ars):
            sr = np.float32(math.sqrt(r))
            hws[j * nr + i] = (mn[0] * sr, mn[1] / sr)
        hws[j * nr + nr - 1] = (np.float32(math.sqrt(float(np.float32(mn[0] * mx[0])))), np.float32(math.sqrt(float(np.float32(mn[1] * mx[1])))))
    step_w = step if step is not None else img_w / W   # :111-118
    step_h = step if step is not None else img_h / H
    xs = linspace_f32(offset[0] * step_w, (offset[0] + W - 1) * step_w, W, use_fma)
    ys = linspace_f32(offset[1] * step_h, (offset[1] + H - 1) * step_h, H, use_fma)
    out = np.empty((H, W, nr * num_branches, 4), np.float32)
    out[..., 0] = xs[None, :, None]
    out[..., 1] = ys[:, None, None]
    out[..., 2] = hws[None, None, :, 0]
    out[..., 3] = hws[None, None, :, 1]
    return out


def mean_average_precision(pred, gt_list, num_classes, iou_threshold=0.5, voc=False):
    """detection/metrics/mean_average_precision.py:10-116 -> (mAP, ap[num_classes] with NaN for classes without ground truth)"""
    pred = _f32(np.asarray(pred, np.float32).reshape(-1, 7))
    stride = int(gt_list[0].shape[1]) if len(gt_list) else 6
    rows, offs = pack_gt(gt_list, stride)
    ap = np.empty(num_classes, np.float32)
    fn = lib().orc_mean_average_precision
    fn.restype = C.c_double
    m = fn(_p(pred), C.c_int64(pred.shape[0]), _p(rows), C.c_int(stride), _p(offs), C.c_int(len(gt_list)), C.c_int(num_classes),
           C.c_float(iou_threshold), C.c_int(int(voc)), _p(ap))
    return float(m), ap
