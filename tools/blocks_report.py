"""Worst deviation of every blocks_small case from the reference's golden, as a multiple of the 1e-4 bar of tests/test_blocks_golden_gpu.py
(|diff| / (1e-4 * (|ref| + max|ref|))), per case and mode.  python3 tools/blocks_report.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import blocks_cases  # noqa: E402
from test_blocks_golden_gpu import MODS  # noqa: E402

g = np.load(os.path.join(REPO, 'tests', 'golden', 'blocks_small.npz'))
for case in sorted(blocks_cases.CASES):
    got = blocks_cases.run_case(case, MODS, torch.device('cuda'))
    worst = {}
    for key in sorted(got):
        if key.endswith('__shape') or key.endswith('__sum_l2') or key.endswith('num_batches_tracked'):
            continue
        ref, val = g[key], got[key]
        scale = float(np.abs(ref).max())
        r = float((np.abs(val.astype(np.float64) - ref) / (1e-4 * (np.abs(ref) + scale) + 1e-12)).max())
        grp = '/'.join(key.split('/')[1:3]) if '/buffers/' not in key else 'buffers'
        if r > worst.get(grp, (0, ''))[0]:
            worst[grp] = (r, key)
    print(case, {k: (round(v[0], 3), v[1].split('/', 2)[-1]) for k, v in worst.items()}, flush=True)
