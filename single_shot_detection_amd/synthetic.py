"""Seeded synthetic inputs for the detection hot path.

Everything here is numpy ``default_rng`` (PCG64) so the very same bytes are produced in the
build container (golden generation against the reference), in the CPU test-suite and on the
GPU box.  No torch RNG is involved: torch's generators differ between CPU and HIP.

Ground-truth row layout follows the reference dataset contract
(``bf/datasets/detection_dataset.py:11-17``): ``[x1, y1, x2, y2, class, score]`` fp32.
"""
import numpy as np

# (input size, C, levels (Cin, H=W, nb)) -- SURVEY.md §8 table (probed shapes of the reference).
CONFIGS = {
    'ssd_mb2_voc': dict(
        size=300, num_classes=21, score_converter='SOFTMAX',
        levels=[(96, 19, 4), (1280, 10, 6), (512, 5, 6), (256, 3, 6), (256, 2, 4), (128, 1, 4)],
        anchor={'type': 'ssd', 'num_scales': 6, 'min_scale': 0.1, 'max_scale': 1.05,
                'aspect_ratios': [[1.0, 2.0]] + [[1.0, 2.0, 3.0]] * 3 + [[1.0, 2.0]] * 2},
        matched=0.5, unmatched=0.5, nms_thr=0.45, loss='ce_hnm'),
    'ssd_300_vgg16_voc': dict(
        size=300, num_classes=81, score_converter='SOFTMAX',
        levels=[(512, 37, 4), (512, 18, 6), (512, 9, 6), (256, 5, 6), (256, 3, 4), (256, 2, 4)],
        anchor={'type': 'ssd', 'num_scales': 6, 'min_scale': 0.15, 'max_scale': 1.05,
                'aspect_ratios': [[1.0, 2.0]] + [[1.0, 2.0, 3.0]] * 3 + [[1.0, 2.0]] * 2},
        matched=0.5, unmatched=0.5, nms_thr=0.45, loss='ce_hnm'),
    'ssd_512_vgg16_coco': dict(
        size=512, num_classes=81, score_converter='SOFTMAX',
        levels=[(512, 64, 4), (512, 32, 6), (512, 16, 6), (256, 8, 6), (256, 4, 6), (256, 2, 4), (256, 1, 4)],
        anchor={'type': 'ssd', 'num_scales': 7, 'min_scale': 0.1, 'max_scale': 1.05,
                'aspect_ratios': [[1.0, 2.0]] + [[1.0, 2.0, 3.0]] * 4 + [[1.0, 2.0]] * 2},
        matched=0.5, unmatched=0.5, nms_thr=0.45, loss='ce_hnm'),
    'retina_rn50_500_coco': dict(
        size=500, num_classes=80, score_converter='SIGMOID',
        levels=[(256, 63, 9), (256, 32, 9), (256, 16, 9), (256, 8, 9), (256, 4, 9)],
        anchor={'type': 'retina_net', 'min_level': 3, 'max_level': 7, 'aspect_ratios': [1.0, 2.0, 0.5],
                'scale': 4.0, 'scales_per_level': 3},
        matched=0.5, unmatched=0.4, nms_thr=0.5, loss='focal_naive'),
    'm2det_512_vgg16_coco': dict(
        size=512, num_classes=81, score_converter='SOFTMAX',
        levels=[(1024, 64, 4), (1024, 32, 6), (1024, 16, 6), (1024, 8, 6), (1024, 4, 4), (1024, 2, 4)],
        anchor={'type': 'ssd', 'num_scales': 6, 'min_scale': 0.07, 'max_scale': 1.05,
                'aspect_ratios': [[1.0, 2.0]] + [[1.0, 2.0, 3.0]] * 3 + [[1.0, 2.0]] * 2},
        matched=0.5, unmatched=0.5, nms_thr=0.45, loss='ce_hnm'),
}


# BASELINE.json calls the SSD-300 workload "VOC-20-class" while the sample file hard-codes num_classes = 81: the
# 21-column variant of the same geometry, for bench.py --config ssd_300_vgg16_voc_c21 (not a golden config)
CONFIGS['ssd_300_vgg16_voc_c21'] = dict(CONFIGS['ssd_300_vgg16_voc'], num_classes=21)


def num_anchors(cfg):
    return sum(h * h * nb for _, h, nb in cfg['levels'])


def make_ground_truth(batch, size, num_classes, seed=1, g_min=1, g_max=8, fixed_g=None, background=True):
    """list[B] of float32 [G_i, 6] rows ``x1,y1,x2,y2,cls,score`` (BASELINE.md §3 'Inputs').

    ``background``: True when class 0 is background (SOFTMAX configs: classes 1..C-1);
    for SIGMOID configs the reference still uses 1-based class ids (``multibox_loss.py:67``),
    so classes are 1..C.
    """
    rng = np.random.default_rng(seed)
    out = []
    hi = num_classes - 1 if background else num_classes
    for _ in range(batch):
        g = fixed_g if fixed_g is not None else int(rng.integers(g_min, g_max + 1))
        xy = rng.uniform(0.0, 0.7 * size, size=(g, 2))
        wh = rng.uniform(0.05, 0.55, size=(g, 2)) * size
        x2y2 = np.minimum(xy + wh, size - 1)
        cls = rng.integers(1, hi + 1, size=(g, 1)).astype(np.float64)
        score = np.ones((g, 1))
        out.append(np.concatenate([xy, x2y2, cls, score], axis=1).astype(np.float32))
    return out


def make_logits(batch, anchors, num_classes, seed=2, trained_like=False):
    """scores ``[B, A*C]`` fp32 N(0,1); ``trained_like`` adds +6 to the background logit."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, anchors, num_classes), dtype=np.float32)
    if trained_like:
        x[..., 0] += 6.0
    return x.reshape(batch, anchors * num_classes)


def make_locs(batch, anchors, seed=3, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((batch, anchors * 4), dtype=np.float32) * np.float32(scale))


def make_feature_maps(batch, levels, seed=23):
    """list[L] of NCHW fp32 N(0,1) source maps at the §8 shapes (head-only microbench input)."""
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((batch, cin, h, h), dtype=np.float32) for cin, h, _ in levels]


def make_map_case(seed, num_images=12, num_classes=6, with_difficult=True, max_gt=6, size=300.0, dup=0.3, noise_fp=2, empty_images=True,
                  unique_scores=True, difficult_p=0.2):
    """Seeded input of detection/metrics/mean_average_precision.py: (predictions [N,7], gts list of [G_i, 6 or 7]).

    Ground truths: random corner boxes, classes 1..num_classes-1, optional difficult column (20 %).  Predictions: a jittered copy of
    ~75 % of the ground truths, duplicates of some of them (a second hit on a matched box is a false positive), and random boxes
    with random classes (incl. classes the image does not contain).  Scores are distinct unless ``unique_scores`` is False."""
    rng = np.random.default_rng(seed)
    gts, preds = [], []
    for i in range(num_images):
        g = int(rng.integers(0 if empty_images else 1, max_gt + 1))
        xy = rng.uniform(0, 0.7 * size, (g, 2))
        wh = rng.uniform(0.05, 0.4, (g, 2)) * size
        box = np.concatenate([xy, np.minimum(xy + wh, size - 1)], 1)
        cls = rng.integers(1, num_classes, (g, 1)).astype(np.float64)
        cols = [box, cls, np.ones((g, 1))]
        if with_difficult:
            cols.append((rng.random((g, 1)) < difficult_p).astype(np.float64))
        gts.append(np.concatenate(cols, 1).astype(np.float32).reshape(g, 7 if with_difficult else 6))
        rows = []
        for k in range(g):
            if rng.random() < 0.75:
                for _ in range(1 + int(rng.random() < dup)):
                    jit = rng.normal(0, 0.06, 4) * np.tile(wh[k], 2)
                    rows.append(np.concatenate([[i], box[k] + jit, cls[k], [rng.uniform(0.2, 1.0)]]))
        for _ in range(int(rng.integers(0, noise_fp + 1))):
            xy2 = rng.uniform(0, 0.7 * size, 2)
            rows.append(np.concatenate([[i], xy2, xy2 + rng.uniform(0.05, 0.4, 2) * size, [rng.integers(1, num_classes)], [rng.uniform(0.01, 0.6)]]))
        if rows:
            preds.append(np.stack(rows))
    pred = np.concatenate(preds, 0).astype(np.float32) if preds else np.zeros((0, 7), np.float32)
    if unique_scores is None:
        pass                                                  # raw scores: a few accidental ties (large benchmark inputs)
    elif unique_scores and pred.shape[0]:
        s = pred[:, 6]
        for _ in range(20):
            if np.unique(s).size == s.size:
                break
            s += rng.uniform(0, 1e-4, s.size).astype(np.float32)
        while np.unique(s).size != s.size:   # tens of thousands of fp32 scores: re-drawing all of them keeps colliding -- move the repeats only
            order = np.argsort(s, kind='stable')
            rep = order[1:][s[order[1:]] == s[order[:-1]]]
            s[rep] = np.nextafter(s[rep], np.float32(2.0))
    elif pred.shape[0]:
        pred[:, 6] = np.round(pred[:, 6] * 20) / 20          # many exact ties
    pred = pred[rng.permutation(pred.shape[0])]
    return np.ascontiguousarray(pred, dtype=np.float32), gts


# golden cases of tests/golden/map.npz: name -> make_map_case kwargs; each is run with voc = False and True
MAP_CASES = {
    'small_diff': dict(seed=11, num_images=12, num_classes=6, with_difficult=True),
    'small_nodiff': dict(seed=12, num_images=12, num_classes=6, with_difficult=False),
    'many_classes': dict(seed=13, num_images=40, num_classes=21, with_difficult=True, max_gt=8, noise_fp=4),
    'dense_dups': dict(seed=14, num_images=6, num_classes=3, with_difficult=True, max_gt=10, dup=0.9, noise_fp=6),
    'no_empty': dict(seed=15, num_images=25, num_classes=81, with_difficult=False, max_gt=5, empty_images=False),
    'few_difficult': dict(seed=20, num_images=30, num_classes=8, with_difficult=True, max_gt=7, difficult_p=0.08),
    'few_difficult2': dict(seed=30, num_images=30, num_classes=8, with_difficult=True, max_gt=7, difficult_p=0.08),
}
