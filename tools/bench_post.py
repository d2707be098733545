#!/usr/bin/env python3
"""Postprocess alone on synthetic logits (for rocprofv3 --kernel-trace --stats):
    python tools/bench_post.py [config] [batch] [worst|trained] [reps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import synthetic as syn
from single_shot_detection_amd.detection.box_coder import BoxCoder
from single_shot_detection_amd.detection.postprocessor import Postprocessor
from single_shot_detection_amd.detection import anchor_generators

name = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
variant = sys.argv[3] if len(sys.argv) > 3 else 'worst'
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = syn.CONFIGS[name]
dev = torch.device('cuda:0')
p = dict(cfg['anchor'])
gens = getattr(anchor_generators, p.pop('type')).build_anchor_generators(**p)
img = torch.empty((1, 3, cfg['size'], cfg['size']), device=dev)
anchors = torch.cat([g.generate(img, (h, h)).reshape(-1) for g, (_, h, _) in zip(gens, cfg['levels'])]).view(-1, 4)
A, C = anchors.shape[0], cfg['num_classes']
softmax = cfg['score_converter'] == 'SOFTMAX'
g = torch.Generator(device=dev).manual_seed(5)
logits = torch.randn((B, A, C), device=dev, generator=g)
if variant.startswith('bg'):   # bg<shift>: background logit + shift (the density knob between "worst" = 0 and "trained" = 6)
    logits[..., 0] += float(variant[2:])
if variant == 'trained':
    if softmax:
        logits[..., 0] += 6.0
    else:
        logits -= 4.6
logits = logits.view(B, -1)
locs = torch.randn((B, A * 4), device=dev, generator=g) * 0.5
post = Postprocessor(BoxCoder(10.0, 5.0), 0.01, {'max_per_class': 100, 'overlap_threshold': cfg['nms_thr']}, cfg['score_converter'], 200)
for _ in range(3):
    post.postprocess_padded((logits, locs), anchors)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    post.postprocess_padded((logits, locs), anchors)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f'{name} b{B} {variant}: {dt * 1e6:.1f} us per call, {B / dt:.0f} img/s, candidates/img {post.last_nms_candidates.sum().item() / B:.0f}')
