"""One postprocess leg, repeated (for rocprofv3 --kernel-trace --stats): python tools/post_leg.py <config> <batch> <trained|worst> [calls]"""
import sys
sys.path.insert(0, '.')
import torch, bench
from single_shot_detection_amd import synthetic as syn
cfg, B, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device('cuda:0')
hp = bench.HotPath(cfg, B, dev)
A, C = hp.anchors.shape[0], syn.CONFIGS[cfg]['num_classes']
softmax = syn.CONFIGS[cfg]['score_converter'] == 'SOFTMAX'
sc = torch.from_numpy(syn.make_logits(B, A, C, seed=2)).to(dev)
locs = torch.from_numpy(syn.make_locs(B, A, seed=3)).to(dev)
if kind == 'trained':
    sc = sc.view(B, A, C)
    if softmax: sc[..., 0] += 6.0
    else: sc -= 6.25
    sc = sc.view(B, -1)
for _ in range(calls):
    hp.post.postprocess_padded((sc, locs), hp.anchors)
torch.cuda.synchronize()
us = bench.gpu_time_us(lambda: hp.post.postprocess_padded((sc, locs), hp.anchors), inner=5)
print('%s b%d %s: %.1f us per call' % (cfg, B, kind, us))
