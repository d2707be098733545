"""Graph-replayed training steps against eager ones, step by step, from the same state.  python3 tools/graph_train_check.py [config] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd.detection import target_assigner as ta  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_mb2_voc'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
A, B = bench.HotPath(cfg, batch, dev), bench.HotPath(cfg, batch, dev)
packed = ta.pack_ground_truth(B.gt, dev)
orig = ta.pack_ground_truth


def params(hp):
    return [p for g in hp.opt.param_groups for p in g['params']]


def diff(tag):
    torch.cuda.synchronize()
    worst = max(float((p.detach() - q.detach()).abs().max() / (q.detach().abs().max() + 1e-12)) for p, q in zip(params(B), params(A)))
    print('%-28s max relative parameter difference %.3e' % (tag, worst))


diff('initial')
for _ in range(3):
    A.train_step()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        B.train_step()
torch.cuda.current_stream().wait_stream(side)
diff('after 3 eager steps each')
ta.pack_ground_truth = lambda gt, device, row=6: packed
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss = B.train_step()
ta.pack_ground_truth = orig
diff('after capture (no step run)')
for k in range(1, 6):
    la = A.train_step()
    graph.replay()
    diff('step %d: loss eager %.5f graph %.5f' % (k, float(la.detach()), float(loss.detach())))
