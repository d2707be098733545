"""Randomised differential check of mean average precision (own radix sort + scans + matching kernels) and of the SSD anchor generator's
constructor modes against the oracle.   python3 tools/stress_metrics.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle  # noqa: E402
from single_shot_detection_amd import synthetic as syn  # noqa: E402
from test_metrics_gpu import _run  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    kw = dict(seed=int(rng.integers(0, 1 << 30)), num_images=int(rng.choice([1, 3, 17, 120, 500])), num_classes=int(rng.choice([2, 4, 21, 81])),
              with_difficult=bool(rng.integers(0, 2)), max_gt=int(rng.choice([1, 4, 12, 40])), noise_fp=int(rng.choice([0, 5, 30])),
              unique_scores=bool(rng.integers(0, 4)))
    if kw['with_difficult']:
        kw['difficult_p'] = float(rng.choice([0.0, 0.05, 0.3]))
    if rng.integers(0, 3) == 0:
        kw['dup'] = float(rng.choice([0.2, 0.8]))
    voc = bool(rng.integers(0, 2))
    try:
        pred, gts = syn.make_map_case(**kw)
        m, ap = _run(pred, gts, kw['num_classes'], voc)
        mo, apo = oracle.mean_average_precision(pred, gts, kw['num_classes'], 0.5, voc)
        # (fp32 sums over thousands of detections of one class in scan order vs the reference's sequential order: a few 1e-6)
        assert (np.isnan(m) and np.isnan(mo)) or abs(m - mo) <= 5e-6, (m, mo)
        np.testing.assert_allclose(ap.numpy(), apo, atol=5e-6, equal_nan=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print('FAIL', dict(case=case, voc=voc, **kw), type(e).__name__, str(e)[:300].replace('\n', ' | '), flush=True)
print('%d mAP cases, %d failures' % (cases, bad))

# ---- SSD anchor generator: every constructor mode, bit for bit against the oracle's restatement (itself pinned by the reference's goldens)
from single_shot_detection_amd.detection.anchor_generators import ssd  # noqa: E402
abad = 0
for case in range(cases):
    ars = [[1.0], [1.0, 2.0], [1.0, 2.0, 3.0], [2.0, 0.5], [1.5, 1.0, 3.0]][int(rng.integers(0, 5))]
    kw = dict(aspect_ratios=ars, flip=bool(rng.integers(0, 2)) and min(ars) >= 1.0, num_branches=int(rng.choice([1, 2, 3])))   # (ratios < 1 with flip: the constructor asserts, like the reference's)
    if rng.integers(0, 2):
        lo = float(rng.uniform(0.05, 0.5))
        kw.update(min_scale=lo, max_scale=lo + float(rng.uniform(0.05, 0.5)))
    else:
        lo = float(rng.uniform(10, 100))
        kw.update(min_size=lo, max_size=lo + float(rng.uniform(5, 150)))
    if rng.integers(0, 2):
        kw['step'] = int(rng.choice([8, 16, 30, 64]))
    if rng.integers(0, 2):
        kw['offset'] = (float(rng.choice([0.0, 0.25, 0.5])), float(rng.choice([0.5, 0.75])))
    img_wh = (int(rng.choice([300, 512, 321])), int(rng.choice([300, 512, 287])))
    fmap_wh = (int(rng.choice([1, 3, 10, 19, 38, 64])), int(rng.choice([1, 2, 10, 19, 38])))
    try:
        gen = ssd.SsdAnchorGenerator(**kw)
        img = torch.empty((1, 3, img_wh[1], img_wh[0]), device='cuda')
        got = gen.generate(img, (fmap_wh[1], fmap_wh[0])).cpu().numpy()
        okw = {k: v for k, v in kw.items() if k != 'aspect_ratios'}
        ref = oracle.ssd_anchor_generator(img_wh, fmap_wh, ars, **okw)
        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), float(np.abs(got - ref).max())
    except Exception as e:   # noqa: BLE001
        abad += 1
        print('FAIL', dict(case=case, img=img_wh, fmap=fmap_wh, **kw), type(e).__name__, str(e)[:300].replace('\n', ' | '), flush=True)
print('%d anchor cases, %d failures' % (cases, abad))
sys.exit(1 if bad or abad else 0)
