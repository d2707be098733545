#!/usr/bin/env python3
"""Event-timed micro-benchmark of the head GEMMs (forward / backward) at a BASELINE config's shapes."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import synthetic as syn  # noqa: E402
from single_shot_detection_amd.detection import detector_builder  # noqa: E402
from single_shot_detection_amd.detection.modules.heads import multi_level_heads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='ssd_300_vgg16_voc')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--fwd-only', action='store_true')
    ap.add_argument('--relu-inputs', action='store_true', help='source maps max(N(0,1), 0) (post-ReLU backbone features: half the operand zeros) instead of N(0,1)')
    ap.add_argument('--cold', type=int, default=0, help='forward only: MB copied between two launches (evicts L2 / the memory-side cache), each launch timed alone')
    ap.add_argument('--sparse', type=float, default=0.0, help='fraction of zero dY rows (per anchor) in the backward input')
    args = ap.parse_args()
    cfg = syn.CONFIGS[args.config]
    levels, C, B = cfg['levels'], cfg['num_classes'], args.batch
    dev = torch.device('cuda')
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).to(dev)
    xs = [torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for x in syn.make_feature_maps(B, levels)]
    if args.relu_inputs:
        xs = [x.detach().clamp_min(0).requires_grad_(True) for x in xs]
    flops = sum(2.0 * h * h * 9 * cin * nb * (C + 4) for cin, h, nb in levels) * B

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_fwd = timed(lambda: multi_level_heads(xs, xs, heads), args.reps)
    if args.cold:
        big = torch.empty(args.cold << 18, dtype=torch.float32, device=dev)
        big2 = torch.empty_like(big)
        ts = []
        with torch.no_grad():
            for _ in range(args.reps):
                big2.copy_(big)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                multi_level_heads(xs, xs, heads)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f'{args.config} B={B}: fwd back to back {t_fwd:.3f} ms | behind a {args.cold} MB copy: median {ts[len(ts) // 2]:.3f} ms, min {ts[0]:.3f}, max {ts[-1]:.3f}')
        return
    if args.fwd_only:
        print(f'{args.config} B={B}: fwd {t_fwd:.3f} ms = {flops / t_fwd / 1e9:.1f} TFLOP/s')
        return
    scores, locs = multi_level_heads(xs, xs, heads)
    gs = torch.randn_like(scores)
    gl = torch.randn_like(locs)
    if args.sparse > 0:
        A = scores.shape[1] // C
        keep = (torch.rand((B, A, 1), device=dev) >= args.sparse).float()
        gs = (gs.view(B, A, C) * keep).view(B, -1)
        gl = (gl.view(B, A, 4) * keep).view(B, -1)

    def bwd():
        for x in xs:
            x.grad = None
        for p in heads.parameters():
            p.grad = None
        s, l = multi_level_heads(xs, xs, heads)
        torch.autograd.backward([s, l], [gs, gl])

    t_fb = timed(bwd, args.reps)
    t_bwd = t_fb - t_fwd
    print(f'{args.config} B={B}: fwd {t_fwd:.3f} ms = {flops / t_fwd / 1e9:.1f} TFLOP/s | '
          f'bwd (dgrad+wgrad+pack) {t_bwd:.3f} ms = {2 * flops / t_bwd / 1e9:.1f} TFLOP/s | fwd+bwd {t_fb:.3f} ms')


if __name__ == '__main__':
    main()
