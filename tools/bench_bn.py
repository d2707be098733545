#!/usr/bin/env python3
"""Event-timed BatchNorm2d (+ReLU) forward and backward on libssdk for maps of the M2Det neck / SSD tail, with the bytes each pass must move:
    python tools/bench_bn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import ops  # noqa: E402

SHAPES = [(16, 512, 64, 64), (16, 896, 64, 64), (16, 128, 64, 64), (16, 256, 32, 32), (16, 256, 16, 16), (16, 256, 4, 4), (32, 256, 19, 19), (32, 512, 10, 10)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device('cuda')
    for B, C, H, W in SHAPES:
        bn = torch.nn.BatchNorm2d(C).to(dev)
        x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        mb = B * C * H * W * 4 / 1e6
        with torch.no_grad():
            fwd = timed(lambda: ops.batch_norm(x, bn, relu=True))
        y = ops.batch_norm(x, bn, relu=True)
        dy = torch.randn_like(y)
        bwd = timed(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
        # forward: statistics read x, apply reads x and writes y (3 passes); backward: statistics read x, y, dy, apply reads x, y, dy and writes dx (7)
        print(f'[{B},{C},{H},{W}] {mb:7.1f} MB: forward (statistics + apply) {fwd:7.1f} us = {3 * mb / fwd:5.2f} TB/s | '
              f'backward (statistics + apply) {bwd:7.1f} us = {7 * mb / bwd:5.2f} TB/s')


if __name__ == '__main__':
    main()
