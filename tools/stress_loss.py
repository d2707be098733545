"""Randomised differential check of target assignment + samplers + MultiboxLoss (forward, in-place target encoding, backward) against the
oracle: random anchor sets, batch sizes, box counts (empty images, duplicates, many boxes), thresholds, class counts, both loss families.
    python3 tools/stress_loss.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle  # noqa: E402
from single_shot_detection_amd.detection.target_assigner import TargetAssigner  # noqa: E402
from test_loss_gpu import check_mask, make_criterion  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    B = int(rng.choice([1, 2, 3, 8, 17]))
    A = int(rng.choice([16, 100, 777, 2268, 8108, 20000]))
    kind = 'ce_hnm' if rng.integers(0, 2) else 'focal'
    Cn = int(rng.choice([2, 5, 21, 81])) if kind == 'ce_hnm' else int(rng.choice([1, 4, 20, 80]))
    size = 300.0
    cxy = rng.uniform(10, size - 10, (A, 2))
    wh = rng.uniform(8, 150, (A, 2))
    anchors = np.concatenate([cxy, wh], 1).astype(np.float32)
    matched = float(rng.choice([0.5, 0.6, 0.35]))
    unmatched = float(rng.choice([matched, matched - 0.1]))
    gt = []
    for i in range(B):
        G = int(rng.choice([0, 1, 2, 5, 9, 40, 140]))
        x1y1 = rng.uniform(0, size - 20, (G, 2))
        bwh = rng.uniform(5, 160, (G, 2))
        cls = rng.integers(1 if kind == 'ce_hnm' else 0, max(Cn, 2) if kind == 'ce_hnm' else Cn, (G, 1)) if G else np.zeros((0, 1))
        rows = np.concatenate([x1y1, np.minimum(x1y1 + bwh, size - 1), cls, np.ones((G, 1))], 1).astype(np.float32)
        if G > 1 and rng.integers(0, 3) == 0:
            rows[1, :4] = rows[0, :4]   # duplicated box
        gt.append(rows)
    tag = dict(case=case, B=B, A=A, kind=kind, C=Cn, matched=matched, unmatched=unmatched, G=[len(g) for g in gt])
    try:
        ref_t, ref_idx = oracle.encode_ground_truth(gt, anchors, matched, unmatched, return_box_idx=True)
        anchors_t = torch.from_numpy(anchors).cuda()
        t, idx = TargetAssigner(matched, unmatched).encode_ground_truth([torch.from_numpy(g) for g in gt], anchors_t, return_box_idx=True)
        assert np.array_equal(idx.cpu().numpy(), ref_idx), 'box_idx'
        assert np.array_equal(t.cpu().numpy().view(np.uint32), ref_t.view(np.uint32)), 'target bits'
        logits = rng.standard_normal((B, A, Cn)).astype(np.float32)
        if kind == 'ce_hnm':
            logits[..., 0] += float(rng.choice([0.0, 3.0]))
        locs = (rng.standard_normal((B, A, 4)) * 0.5).astype(np.float32)
        okind = 'ce' if kind == 'ce_hnm' else 'focal'
        if okind == 'ce':
            ref_mask, bg = oracle.hard_negative_mining(logits.reshape(B, -1), ref_t, 3, 5, return_bgloss=True)
        else:
            ref_mask, bg = oracle.naive_sampler(logits.reshape(B, -1), ref_t), None
        scores_t = torch.from_numpy(logits.reshape(B, -1)).cuda().requires_grad_(True)
        locs_t = torch.from_numpy(locs.reshape(B, -1)).cuda().requires_grad_(True)
        crit = make_criterion(kind)
        loss, class_loss, loc_loss = crit((scores_t, locs_t), anchors_t, t)
        (2.0 * class_loss + 0.5 * loc_loss).backward()
        mask = crit.last_sampled_mask.cpu().numpy().astype(bool)
        check_mask(mask, ref_mask, bg)
        tgt_ref = ref_t.copy()
        vals, ds, dl = oracle.multibox_loss(logits.reshape(B, -1), locs.reshape(B, -1), anchors, tgt_ref, mask, kind=okind, reduce_mean=True)
        got = np.array([loss.item(), class_loss.item(), loc_loss.item()])
        np.testing.assert_allclose(got, vals, rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(scores_t.grad.view(B, A, Cn).cpu().numpy(), 2.0 * ds, rtol=2e-4, atol=2e-7)
        np.testing.assert_allclose(locs_t.grad.view(B, A, 4).cpu().numpy(), 0.5 * dl, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(t.cpu().numpy(), tgt_ref, rtol=1e-6, atol=2e-5)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print('FAIL', tag, type(e).__name__, str(e)[:400].replace('\n', ' | '), flush=True)
print('%d cases, %d failures' % (cases, bad))
sys.exit(1 if bad else 0)
