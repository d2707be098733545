#!/usr/bin/env python3
"""Kernel lists of (a) detection.init(graph_hot_path=True)'s step_fn and (b) bench.HotPath's whole-step graph, for one configuration:
    rocprofv3 --kernel-trace --stats -d out -o p -- python3 tools/stepfn_trace.py stepfn|hotpath [ncls] [batch]
(the backbone's kernels show up in (a) as well: compare the ssdk:: rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import synthetic as syn  # noqa: E402

mode = sys.argv[1]
ncls = int(sys.argv[2]) if len(sys.argv) > 2 else 21
b = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device('cuda:0')
if mode == 'stepfn':
    from single_shot_detection_amd.detection import init as det_init
    cfg = syn.CONFIGS['ssd_300_vgg16_voc']
    model = {'base': {'name': 'torchvision_vgg16_bn', 'pretrained': False},
             'detector': {'num_classes': ncls, 'use_depthwise': False, 'features': {'name': 'Features', 'out_layers': (32, 42), 'last_feature_layer': 42},
                          'extras': {'layers': (('s', 512), ('s', 256), ('s', 256), ('s', 256))}},
             'anchor_generator': dict(cfg['anchor'])}
    args = ({'xy_scale': 10.0, 'wh_scale': 5.0},
            {'score_threshold': .01, 'max_total': 200, 'nms': {'max_per_class': 100, 'overlap_threshold': .45}, 'score_converter': 'SOFTMAX'},
            {'classification_loss': {'name': 'CrossEntropyLoss'}, 'localization_loss': {'name': 'SmoothL1Loss'}, 'classification_weight': 1.0, 'localization_weight': 1.0},
            {'name': 'hard_negative_mining', 'negative_per_positive_ratio': 3, 'min_negative_per_image': 5}, {'matched_threshold': 0.5, 'unmatched_threshold': 0.5})
    wrapper, init_state, step_fn = det_init.init(dev, model, *args, graph_hot_path=True)
    det = wrapper.model
    det.train()
    hot = [p for n, p in det.predictor.named_parameters() if not n.startswith('features.')]
    opt = torch.optim.SGD(hot, lr=1e-4, momentum=0.9, fused=True)
    imgs = torch.randn((b, 3, 300, 300), device=dev)
    import numpy as np
    gt = [torch.from_numpy(g) for g in syn.make_ground_truth(b, 300, ncls, seed=1)]
    st = init_state()
    for k in range(10):
        opt.zero_grad(set_to_none=True)
        det.zero_grad(set_to_none=True)
        loss, _, st = step_fn(k, 'train', (imgs, gt), st)
        loss.backward()
        opt.step()
else:
    from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth
    from single_shot_detection_amd.graphs import GraphedCallable
    hp = bench.HotPath('ssd_300_vgg16_voc_c21' if ncls == 21 else 'ssd_300_vgg16_voc', b, dev)
    hp.gt = PackedGroundTruth.from_list(hp.gt, dev)
    g = GraphedCallable(hp.train_step, [])
    for _ in range(10):
        g()
torch.cuda.synchronize()
