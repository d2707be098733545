"""Which part of the step is not replay-safe?  Captures (a) forward + loss, (b) + backward, without the optimizer: with constant parameters
every replay must reproduce the eager values.   python3 tools/graph_bisect.py [config] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd.detection import target_assigner as ta  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_mb2_voc'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
hp = bench.HotPath(cfg, batch, dev)
packed = ta.pack_ground_truth(hp.gt, dev)
ta.pack_ground_truth = lambda gt, device, row=6: packed
params = [p for g in hp.opt.param_groups for p in g['params']]


def fwd():
    scores, locs = hp.forward_heads()
    target = hp.assigner.encode_ground_truth(hp.gt, hp.anchors)
    loss, cl, ll = hp.criterion((scores, locs), hp.anchors, target)
    return loss, scores, locs, target


def fwd_bwd():
    for p in params:
        p.grad = None
    for s in hp.inputs:
        s.grad = None
    loss, scores, locs, target = fwd()
    loss.backward()
    return loss, scores, locs, target


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def summary(out, grads=False):
    loss, scores, locs, target = out
    v = [float(loss.detach()), float(scores.detach().double().abs().sum()), float(locs.detach().double().abs().sum()), float(target.double().abs().sum())]
    if grads:
        v += [float(sum(p.grad.double().abs().sum() for p in params)), float(sum(s.grad.double().abs().sum() for s in hp.inputs if s.grad is not None))]
    return ['%.6e' % x for x in v]


print('eager fwd      ', summary(fwd()))
g, out = capture(fwd)
for k in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('replay %d fwd   ' % k, summary(out))
print('eager fwd+bwd  ', summary(fwd_bwd(), True))
g2, out2 = capture(fwd_bwd)
for k in range(3):
    g2.replay()
    torch.cuda.synchronize()
    print('replay %d f+b   ' % k, summary(out2, True))
