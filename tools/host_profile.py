"""Where does the host spend a training step?  cProfile of bench.HotPath.train_step (GPU work is asynchronous: what is listed is enqueue
time).   python3 tools/host_profile.py [config] [batch] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_mb2_voc'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
hp = bench.HotPath(cfg, batch, torch.device('cuda:0'))
for _ in range(5):
    hp.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    hp.train_step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%s b%d: host enqueue %.3f ms/step, with the GPU drained %.3f ms/step' % (cfg, batch, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    hp.train_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
