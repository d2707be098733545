#!/usr/bin/env python3
"""Calibration of the CPU restatement (oracle/, what bench.py's `cpu_baseline` times on the GPU box) against the REFERENCE ITSELF, in the
build container where /root/reference is importable: the same seeded inputs through the reference's
TargetAssigner.encode_ground_truth and MultiboxLoss (forward + backward) and through the oracle's C functions, wall-clock, same
thread count.  Prints a markdown table with the factor reference / oracle per stage (BASELINE.md section 4 records it).
    python tools/cpu_calibration.py [config] [batch]"""
import functools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg   # noqa: E402  (installs the import shims and imports the reference's modules)
import oracle  # noqa: E402
from single_shot_detection_amd import synthetic as syn  # noqa: E402


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), float(np.mean(ts))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    cfg = syn.CONFIGS[name]
    C = cfg['num_classes']
    threads = torch.get_num_threads()
    anchors = gg.ref_anchors(cfg)
    A = anchors.shape[0]
    gt = syn.make_ground_truth(B, cfg['size'], C, seed=1, fixed_g=8, background=cfg['score_converter'] == 'SOFTMAX')
    logits = syn.make_logits(B, A, C, seed=2)
    locs = syn.make_locs(B, A, seed=3, scale=0.5)
    anchors_np = anchors.numpy()
    rows = []
    # T1-T3: target assign
    ta = gg.TargetAssigner(cfg['matched'], cfg['unmatched'])
    gt_t = [torch.from_numpy(g) for g in gt]
    r_ref = best(lambda: ta.encode_ground_truth(gt_t, anchors))
    r_orc = best(lambda: oracle.encode_ground_truth(gt, anchors_np, cfg['matched'], cfg['unmatched']))
    rows.append(('target assign (encode_ground_truth)', r_ref, r_orc))
    # S1 + L1-L3: sampler + loss forward + backward
    if cfg['loss'] == 'ce_hnm':
        smp = functools.partial(gg.sampler.hard_negative_mining, negative_per_positive_ratio=3, min_negative_per_image=5)
        cl, kind = {'name': 'CrossEntropyLoss'}, 'ce'
    else:
        smp, cl, kind = gg.sampler.naive_sampler, {'name': 'SigmoidFocalLoss', 'gamma': 2.0, 'alpha': 0.25}, 'focal'
    crit = gg.MultiboxLoss(sampler=smp, box_coder=gg.BoxCoder(10.0, 5.0), classification_loss=cl, localization_loss={'name': 'SmoothL1Loss'})
    target0 = ta.encode_ground_truth(gt_t, anchors)

    def ref_loss():
        s = torch.from_numpy(logits).requires_grad_(True)
        l = torch.from_numpy(locs).requires_grad_(True)
        loss = crit((s, l), anchors, target0.clone())[0]
        loss.backward()
    target_np = oracle.encode_ground_truth(gt, anchors_np, cfg['matched'], cfg['unmatched'])

    def orc_loss():
        if kind == 'ce':
            mask = oracle.hard_negative_mining(logits, target_np, 3, 5)
        else:
            mask = oracle.naive_sampler(logits, target_np)
        oracle.multibox_loss(logits, locs, anchors_np, target_np.copy(), mask, kind=kind)
    rows.append(('sampler + loss forward + backward', best(ref_loss), best(orc_loss)))
    print(f'config {name}, B = {B}, A = {A}, C = {C}, G = 8, {threads} torch threads / {oracle.max_threads()} OpenMP threads, {os.cpu_count()} cores')
    print('| stage | reference (best / mean of 3) | oracle (best / mean of 3) | reference / oracle |')
    print('|---|---|---|---|')
    for what, (rb, rm), (ob, om) in rows:
        print(f'| {what} | {rb * 1e3:.1f} / {rm * 1e3:.1f} ms | {ob * 1e3:.1f} / {om * 1e3:.1f} ms | {rb / ob:.1f}x |')


if __name__ == '__main__':
    main()
