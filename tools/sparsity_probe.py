#!/usr/bin/env python3
"""How sparse is the loss gradient the head backward sees in the bench (per level: pixel rows / anchors with a non-zero gradient)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
hp = bench.HotPath(sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc', 32, torch.device('cuda:0'))
scores, locs = hp.forward_heads()
s = scores.detach().requires_grad_(True)
l = locs.detach().requires_grad_(True)
target = hp.assigner.encode_ground_truth(hp.gt, hp.anchors)
loss, _, _ = hp.criterion((s, l), hp.anchors, target)
loss.backward()
B, C = 32, hp.C
gs = s.grad.view(B, -1, C)
nz = (gs != 0).any(dim=2) | (l.grad.view(B, -1, 4) != 0).any(dim=2)     # [B, A]
off = 0
for cin, h, nb in hp.levels:
    a = h * h * nb
    m = nz[:, off:off + a].view(B, h * h, nb)
    print(f'level {h}x{h} nb={nb}: anchors with grad {int(m.sum())} of {B * a} ({100 * m.float().mean():.2f} %), pixel rows {int(m.any(2).sum())} of {B * h * h} '
          f'({100 * m.any(2).float().mean():.2f} %), per anchor type {m.sum((0, 1)).tolist()}')
    off += a
