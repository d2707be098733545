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
        assert (np.isnan(m) and np.isnan(mo)) or abs(m - mo) <= 2e-6, (m, mo)
        np.testing.assert_allclose(ap.numpy(), apo, atol=2e-6, equal_nan=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print('FAIL', dict(case=case, voc=voc, **kw), type(e).__name__, str(e)[:300].replace('\n', ' | '), flush=True)
print('%d mAP cases, %d failures' % (cases, bad))
sys.exit(1 if bad else 0)
