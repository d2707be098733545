"""Head GEMM forward, level by level: which level of a config keeps the grouped launch below the fp32 MFMA peak?
    python3 tools/heads_level_probe.py [config] [batch]
Every level alone through ssdk_heads_fwd (the grouped launch of ONE level), then all together; TFLOP/s of the algorithmic work."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import bench  # noqa: E402
from single_shot_detection_amd import synthetic as syn  # noqa: E402
from single_shot_detection_amd.detection.modules.heads import multi_level_heads  # noqa: E402
from test_heads_gpu import build_heads  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc_c21'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    cfg = syn.CONFIGS[name]
    C = cfg['num_classes']
    levels = [(cin, h, nb) for cin, h, nb in cfg['levels']]
    rng = np.random.default_rng(1)
    for sel in [[i] for i in range(len(levels))] + [list(range(len(levels)))]:
        lv = [levels[i] for i in sel]
        weights = {}
        for i, (cin, h, nb) in enumerate(lv):
            for k, nout in (('score', nb * C), ('loc', nb * 4)):
                weights[(k, i)] = (rng.standard_normal((nout, cin, 3, 3), dtype=np.float32) * np.float32(0.02), np.zeros((nout,), np.float32))
        heads = build_heads(lv, C, weights)
        xs = [torch.randn(B, cin, h, h, device='cuda').contiguous(memory_format=torch.channels_last) for cin, h, nb in lv]
        with torch.no_grad():
            us = bench.gpu_time_us(lambda: multi_level_heads(xs, xs, heads), inner=5, reps=5)
        fl = bench.head_flops_per_image(lv, C) * B
        print('%-28s levels %-14s N %-16s %8.1f us  %6.1f TFLOP/s  %.3f of peak' % (
            name + ' b%d' % B, sel, [nb * C + nb * 4 for _, _, nb in lv], us, fl / us / 1e6, fl / us / 1e6 / 157.3), flush=True)


if __name__ == '__main__':
    main()
