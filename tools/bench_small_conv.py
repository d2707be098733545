#!/usr/bin/env python3
"""Event-timed forward of the SSD-300 extras' convolutions one by one (batch 32): back-to-back launches of the same layer
("warm") and the same with a 256 MB copy between launches ("cold": caches and the code's L2 lines evicted)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import ops  # noqa: E402

LAYERS = [  # cin, cout, k, stride, pad, hin
    (1024, 256, 1, 1, 0, 19), (256, 512, 3, 2, 1, 19), (512, 128, 1, 1, 0, 10), (128, 256, 3, 2, 1, 10),
    (256, 128, 1, 1, 0, 5), (128, 256, 3, 1, 0, 5), (256, 128, 1, 1, 0, 3), (128, 256, 3, 1, 0, 3)]
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
big = torch.empty(64 << 20, dtype=torch.float32, device=dev)
big2 = torch.empty_like(big)
for cin, cout, k, s, p, h in LAYERS:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device=dev).contiguous(memory_format=torch.channels_last) * 0.01
    with torch.no_grad():
        for _ in range(3): ops.conv2d(x, w, None, s, p)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        for _ in range(20): ops.conv2d(x, w, None, s, p)
        e[1].record()
        torch.cuda.synchronize()
        warm = e[0].elapsed_time(e[1]) / 20 * 1e3
        cold = []
        for _ in range(5):
            big2.copy_(big)
            e[2].record()
            ops.conv2d(x, w, None, s, p)
            e[3].record()
            torch.cuda.synchronize()
            cold.append(e[2].elapsed_time(e[3]) * 1e3)
    # backward-data only (the weight does not ask for a gradient): one re-layout up front, as a prepared step does
    xg = x.clone().requires_grad_(True)
    y = ops.conv2d(xg, w, None, s, p)
    dy = torch.randn_like(y)
    for _ in range(3): torch.autograd.grad(y, xg, dy, retain_graph=True)
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(20): torch.autograd.grad(y, xg, dy, retain_graph=True)
    e[1].record()
    torch.cuda.synchronize()
    bwd = e[0].elapsed_time(e[1]) / 20 * 1e3
    flops = 2.0 * B * ((h + 2 * p - k) // s + 1) ** 2 * cin * k * k * cout
    print(f'{cin:5d}->{cout:4d} k{k} s{s} {h:2d}x{h:<2d}: warm {warm:6.1f} us/launch ({flops / warm / 1e6:6.1f} TFLOP/s)   cold {min(cold):6.1f}..{max(cold):6.1f} us   dgrad (incl. re-layout) {bwd:6.1f} us')
