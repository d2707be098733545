#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one bench train step (no GPU wait) vs the step time?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
hp = bench.HotPath('ssd_300_vgg16_voc', int(sys.argv[1]) if len(sys.argv) > 1 else 32, torch.device('cuda:0'))
for _ in range(5):
    hp.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    hp.train_step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
