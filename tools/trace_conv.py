#!/usr/bin/env python3
"""Experiment: per-slice cycle stamps of the head GEMM (needs a library built with -DSSDK_CONV_TRACE)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import _lib, synthetic as syn  # noqa: E402
from single_shot_detection_amd.detection import detector_builder  # noqa: E402
from single_shot_detection_amd.detection.modules.heads import multi_level_heads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = syn.CONFIGS['ssd_300_vgg16_voc']
levels, C = cfg['levels'], cfg['num_classes']
dev = torch.device('cuda')
heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).to(dev)
xs = [torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last) for x in syn.make_feature_maps(B, levels)]
with torch.no_grad():
    for _ in range(3):
        multi_level_heads(xs, xs, heads)
torch.cuda.synchronize()
NB = 32
n = NB * 4 * 64 * 8
buf = (ctypes.c_ulonglong * n)()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.ssdk_debug_read_trace(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 4, 64, 8).astype(np.int64)
for blk in range(NB):
    for w in range(1):
        top = t[blk, w, :, 0]
        if top[0] == 0:
            continue
        hw = int(t[blk, w, 0, 7])
        per = np.diff(top[4:60])
        seg = [np.mean(t[blk, w, 4:59, i + 1] - t[blk, w, 4:59, i]) for i in range(5)]
        seg[0] = np.mean(t[blk, w, 4:59, 1] - t[blk, w, 4:59, 6]); dma = np.mean(t[blk, w, 4:59, 6] - t[blk, w, 4:59, 0])
        print(f'blk {blk:2d} xcc {(hw >> 32) & 15} simd {(hw >> 4) & 3} slot {hw & 15} tn {(hw >> 40) & 15}: cycles/slice {per.mean():7.0f} | issue {dma:5.0f} gk0 {seg[0]:6.0f} gk1 {seg[1]:6.0f} '
              f'gk2 {seg[2]:6.0f} gk3 {seg[3]:6.0f} barrier {seg[4]:6.0f} next-top {np.mean(t[blk, w, 5:60, 0] - t[blk, w, 4:59, 5]):5.0f}')
print('slice stamps of blk 0 wave 0 (top deltas):', np.diff(t[0, 0, 4:40, 0]).tolist())
