#!/usr/bin/env python3
"""N replays of the HIP-graph-captured training step (bench.HotPath.train_step) for a rocprofv3 --kernel-trace run: under the tracer the
host cannot keep an eager step's queue full, a replay shows what the GPU does with the streams when it can.
    SSDK_MAIN_WGS=448 SSDK_SIDE_WGS=64 python tools/graph_step_trace.py [config] [batch] [replays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth  # noqa: E402
from single_shot_detection_amd.graphs import GraphedCallable  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda:0')
hp = bench.HotPath(cfg, batch, dev)
hp.gt = PackedGroundTruth.from_list(hp.gt, dev, capacity=sum(len(g) for g in hp.gt) + 7)
g = GraphedCallable(hp.train_step, [], warmup=2)
for _ in range(n):
    g()
torch.cuda.synchronize()
print('loss', float(g.static_out.detach()))
