#!/usr/bin/env python3
"""Where does the HOST spend its time when it enqueues one bench train step?   python3 tools/host_profile.py [config] [batch] [top]
Enqueue time per step (no GPU wait) against the step time, then cProfile of 30 steps sorted by own time."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc_c21'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
hp = bench.HotPath(cfg, batch, torch.device('cuda:0'))
for _ in range(5):
    hp.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    hp.train_step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%s b%d: enqueue %.3f ms/step, total %.3f ms/step' % (cfg, batch, (t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    hp.train_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(top)
