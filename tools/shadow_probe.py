#!/usr/bin/env python3
"""Does work that needs no LDS slot run in the shadow of the persistent head GEMM?  The forward head GEMM (stream-K, 512 workgroups
holding both 64 KB LDS slots of every CU for ~1.6 ms) on the main stream; on a second stream, started at the same time, what a training
step could move there because it does not depend on the forward pass: the zero-fill of the heads' data-gradient buffers (150 MB at
SSD-300 batch 32) and the weight re-layouts of the backward GEMMs (4 KB of LDS per workgroup).
    python tools/shadow_probe.py [config] [batch]
Prints: GEMM alone, side work alone, both started together (time until both are done)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import ops, synthetic as syn  # noqa: E402
from single_shot_detection_amd.detection import detector_builder  # noqa: E402
from single_shot_detection_amd.detection.modules.heads import multi_level_heads  # noqa: E402


def main():
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    dev = torch.device('cuda')
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).to(dev)
    xs = [torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last) for x in syn.make_feature_maps(B, levels)]
    dxs = [torch.empty_like(x) for x in xs]
    convs = torch.nn.ModuleList([torch.nn.Conv2d(cin, 4 * (C + 4), 3, padding=1) for cin, _, _ in levels]).to(dev)
    for m in convs:
        m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def gemm():
        with torch.no_grad():
            multi_level_heads(xs, xs, heads)

    def side_work():
        for d in dxs:
            d.zero_()
        ops.prepare_weight_transposes(convs)

    def timed(mode, reps=40):
        """mode 'gemm': the GEMM only; 'serial': side work, then the GEMM, one stream; 'shadow': the side work on the second stream, forked in
        front of the GEMM and joined behind it.  Back to back, one synchronisation at the end."""
        def once():
            if mode == 'serial':
                side_work()
                gemm()
            elif mode == 'shadow':
                side.wait_stream(main)
                gemm()
                with torch.cuda.stream(side):
                    side_work()
                main.wait_stream(side)
            else:
                gemm()
        for _ in range(5):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps):
            once()
        e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    with torch.enable_grad():
        side_bytes = sum(d.numel() * 4 for d in dxs)
        for rnd in range(3):
            a, b, c = timed('gemm'), timed('serial'), timed('shadow')
            print(f'{cfg_name} B={B}: head GEMM {a:.1f} us | + {side_bytes / 1e6:.0f} MB zero-fill + {len(levels)} weight re-layouts on the same stream {b:.1f} us (+{b - a:.1f}) | '
                  f'the same on a second stream, forked in front of the GEMM and joined behind it {c:.1f} us (+{c - a:.1f})')


if __name__ == '__main__':
    main()
