#!/usr/bin/env python3
"""Which decomposition of a pyramid-tail convolution is fastest?  Forward launches (incl. the zero-fill of a split-K output) of the SSD
extras' layers under forced (column blocks, K splits) settings (SSDK_CONV_FORCE), event-timed over back-to-back calls.
    python3 tools/conv_decomp_sweep.py [batch] [config: ssd300 | ssd512]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = sys.argv[2] if len(sys.argv) > 2 else 'ssd300'
LAYERS = {'ssd300': [(512, 256, 1, 1, 0, 18), (256, 512, 3, 2, 1, 18), (512, 128, 1, 1, 0, 9), (128, 256, 3, 2, 1, 9), (256, 128, 1, 1, 0, 5),
                     (128, 256, 3, 2, 1, 5), (256, 128, 1, 1, 0, 3), (128, 256, 3, 2, 1, 3)],
          'ssd512': [(512, 256, 1, 1, 0, 32), (256, 512, 3, 2, 1, 32), (512, 128, 1, 1, 0, 16), (128, 256, 3, 2, 1, 16), (256, 128, 1, 1, 0, 8),
                     (128, 256, 3, 2, 1, 8)],
          # MLFPN neck of m2det_512_vgg16_coco (bf/modules/features.py:215-393): reducers, TUM down / up / smooth layers at their map sizes
          'm2det': [(512, 512, 1, 1, 0, 64), (1024, 256, 1, 1, 0, 32), (768, 128, 1, 1, 0, 64), (768, 256, 3, 2, 1, 64), (256, 256, 3, 2, 1, 64),
                    (256, 256, 3, 2, 1, 32), (256, 256, 3, 2, 1, 16), (256, 256, 3, 2, 1, 8), (256, 256, 3, 2, 1, 4), (256, 768, 1, 1, 0, 32),
                    (256, 256, 1, 1, 0, 32), (256, 256, 1, 1, 0, 16), (256, 256, 1, 1, 0, 8), (256, 256, 1, 1, 0, 4), (256, 256, 1, 1, 0, 2),
                    (256, 128, 1, 1, 0, 64), (256, 128, 1, 1, 0, 32), (256, 128, 1, 1, 0, 16), (256, 128, 1, 1, 0, 8), (256, 128, 1, 1, 0, 4),
                    (256, 128, 1, 1, 0, 2)]}[cfg]
dev = torch.device('cuda')


def timed(x, w, s, p, reps=30):
    with torch.no_grad():
        for _ in range(3):
            ops.conv2d(x, w, None, s, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, w, None, s, p)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cin, cout, k, s, p, h in LAYERS:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.01).contiguous(memory_format=torch.channels_last)
    ho = (h + 2 * p - k) // s + 1
    flops = 2.0 * B * ho * ho * cin * k * k * cout
    os.environ.pop('SSDK_CONV_FORCE', None)
    base = timed(x, w, s, p)
    res = []
    tiles_n = (cout + 31) // 32
    slices = k * k * cin // 32
    for nb in ([] if 'nosweep' in sys.argv else sorted({1, 2, 4, 8, 16} & set(range(1, tiles_n + 1)) | {tiles_n})):
        if nb < (tiles_n + 3) // 4:
            continue
        for ks in (1, 2, 3, 4, 6, 8, 12):
            if ks > 1 and slices // ks < 2:
                continue
            os.environ['SSDK_CONV_FORCE'] = f'{nb},{ks}'
            res.append((timed(x, w, s, p), nb, ks))
    os.environ.pop('SSDK_CONV_FORCE', None)
    res.sort()
    m = B * ho * ho
    print(f'{cin:4d}->{cout:3d} k{k} s{s} {h:2d}->{ho:2d} (M {m:6d}, m_tiles {(m + 127) // 128:3d}, slices {slices:3d}): default {base:6.1f} us ({flops / base / 1e6:5.1f} TF/s) | best '
          + '  '.join(f'nb{nb} ks{ks} {t:5.1f}' for t, nb, ks in res[:4]), flush=True)
