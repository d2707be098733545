"""Randomised differential check of the multi-level heads (forward, dense and sparse backward) against torch's fp32 CPU convolution:
random level lists (channels incl. Cin % 32 != 0, map sizes 1 .. 40, anchor counts, class counts), batch sizes that hit the split-K,
the multi-cut stream-K and the whole-tile forms, gradient densities and forced backward modes; some cases as two single-head calls.   python3 tools/stress_heads.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_heads_gpu import _heads_vs_torch_cpu_conv, _single_heads_vs_torch_cpu_conv  # noqa: E402


class Env(object):
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n_levels = int(rng.integers(1, 5))
    levels = []
    for _ in range(n_levels):
        cin = int(rng.choice([24, 32, 64, 96, 128, 256, 512]))
        h = int(rng.choice([1, 2, 3, 5, 9, 10, 19, 38]))
        if cin >= 256 and h > 19:
            h = 19
        levels.append((cin, h, int(rng.choice([1, 3, 4, 6, 9]))))
    C = int(rng.choice([2, 5, 21, 81]))
    B = int(rng.choice([1, 2, 3, 8, 16]))
    work = sum(B * h * h * 9 * cin * nb * (C + 4) for cin, h, nb in levels)
    if work > 6e9:   # (the CPU reference does three such convolutions)
        B = 1
    density, mode = [(1.0, None), (0.05, None), (0.05, '1'), (0.05, '2'), (0.4, '0'), (0.4, '1'), (0.4, '2')][int(rng.integers(0, 7))]
    tag = dict(case=case, levels=levels, C=C, B=B, density=density, mode=mode)
    try:
        if mode is None and case % 4 == 3:   # every fourth unforced case as two single-head calls (score and loc towers apart: round 5)
            tag['single_heads'] = True
            os.environ.pop('SSDK_HEADS_BWD_MODE', None)
            _single_heads_vs_torch_cpu_conv(levels, C, B, density, bool(case & 4), seed=case)
        else:
            _heads_vs_torch_cpu_conv(levels, C, B, density, mode, Env())
    except Exception as e:   # noqa: BLE001
        bad += 1
        print('FAIL', tag, type(e).__name__, str(e)[:400].replace('\n', ' | '), flush=True)
    if case % 10 == 9:
        print('... %d cases done' % (case + 1), flush=True)
os.environ.pop('SSDK_HEADS_BWD_MODE', None)
print('%d cases, %d failures' % (cases, bad))
sys.exit(1 if bad else 0)
