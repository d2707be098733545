#!/usr/bin/env python3
"""Deterministic mode, localised: two HotPaths from one state run the same training step; every pyramid source map, its gradient, the
input gradients, the loss and every parameter gradient are compared bit for bit and the mismatches are named.
    python tools/determinism_step_diag.py [config] [batch] [0|1 deterministic, default 1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import ops  # noqa: E402


def state(hp):
    mods = [m for m in (hp.heads, hp.extras, hp.tower, hp.neck) if m is not None]
    out = [p for p in hp.params]
    out += [b for m in mods for n, b in m.named_buffers() if not n.startswith('base.')]
    return out


def run(hp):
    rec = {}
    orig = hp.pyramid

    def pyramid():
        srcs = orig()
        for i, t in enumerate(srcs):
            if t.requires_grad and not t.is_leaf:
                t.retain_grad()
        rec['sources'] = srcs
        return srcs
    hp.pyramid = pyramid
    hp.overlap = False
    # what the heads' backward produced (dx per level, before autograd adds the tail's share) and every Conv2dBn block's own dx / dy
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    from single_shot_detection_amd.bf.modules.conv import Conv2dBn
    orig_lb = heads_mod._launch_backward

    def lb(ctx, dscores, dlocs, needs, s_tot, l_tot):
        out = orig_lb(ctx, dscores, dlocs, needs, s_tot, l_tot)
        rec['heads_dscores'] = dscores.clone()
        rec['heads_dx'] = [None if t is None else t.clone() for t in out[0::5]]
        return out
    heads_mod._launch_backward = lb
    hooks = []
    rec['blocks'] = {}
    for root, tag in ((hp.extras, 'extras'), (hp.tower, 'tower'), (hp.neck, 'neck')):
        if root is None:
            continue
        for name, m in root.named_modules():
            if isinstance(m, Conv2dBn):
                def hook(mod, gin, gout, key=tag + '.' + name):
                    rec['blocks'][key] = (None if gin[0] is None else gin[0].clone(), gout[0].clone())
                hooks.append(m.register_full_backward_hook(hook))
    try:
        loss = hp.train_step()
        torch.cuda.synchronize()
    finally:
        heads_mod._launch_backward = orig_lb
        for h in hooks:
            h.remove()
    hp.pyramid = orig
    rec['loss'] = loss.detach().clone()
    return rec


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    det = (sys.argv[3] if len(sys.argv) > 3 else '1') == '1'
    dev = torch.device('cuda:0')
    ops.set_deterministic(det)
    a, b = bench.HotPath(cfg, batch, dev), bench.HotPath(cfg, batch, dev)
    a.train_step()
    b.train_step()
    with torch.no_grad():
        for s, d in zip(state(a), state(b)):
            d.copy_(s)
        for p, q in zip(a.params, b.params):
            b.opt.state[q]['momentum_buffer'].copy_(a.opt.state[p]['momentum_buffer'])
    for rnd in range(2):
        ra, rb = run(a), run(b)
        bad = 0

        def cmp(name, x, y):
            nonlocal bad
            if x is None or y is None:
                print(f'  {name}: missing ({x is None}, {y is None})')
                return
            if not torch.equal(x, y):
                bad += 1
                d = (x - y).abs()
                print(f'  MISMATCH {name} {tuple(x.shape)}: max |diff| {float(d.max()):.3e} on scale {float(x.abs().max()):.3e}, {int((d > 0).sum())} of {x.numel()} elements')
        print(f'round {rnd}: loss {float(ra["loss"])!r} / {float(rb["loss"])!r}')
        cmp('loss', ra['loss'], rb['loss'])
        for i, (x, y) in enumerate(zip(ra['sources'], rb['sources'])):
            cmp(f'source[{i}] forward', x.detach(), y.detach())
        for i, (x, y) in enumerate(zip(ra['sources'], rb['sources'])):
            cmp(f'source[{i}] gradient', x.grad, y.grad)
        cmp('heads dscores', ra['heads_dscores'], rb['heads_dscores'])
        for i, (x, y) in enumerate(zip(ra['heads_dx'], rb['heads_dx'])):
            if x is not None:
                cmp(f'heads dx[{i}]', x, y)
        for key in ra['blocks']:
            cmp(f'{key} dy (gradient of the block output)', ra['blocks'][key][1], rb['blocks'][key][1])
            if ra['blocks'][key][0] is not None:
                cmp(f'{key} dx', ra['blocks'][key][0], rb['blocks'][key][0])
        names = {id(p): n for m, tag in ((a.heads, 'heads'), (a.extras, 'extras'), (a.tower, 'tower'), (a.neck, 'neck')) if m is not None for n, p in ((tag + '.' + k, v) for k, v in m.named_parameters())}
        for p, q in zip(a.params, b.params):
            cmp('grad ' + names.get(id(p), '?'), p.grad, q.grad)
        print(f'  -> {bad} mismatching tensors')


if __name__ == '__main__':
    main()
