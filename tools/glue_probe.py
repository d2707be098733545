#!/usr/bin/env python3
"""Which torch-side (non-libssdk) device operations does a training step of a config run, and from where?  torch.profiler over a few
steps of bench.HotPath: per aten op that launches a kernel, the call count per step and the innermost frames of this repository.
    python tools/glue_probe.py [config] [batch]"""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'm2det_512_vgg16_coco'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    hp = bench.HotPath(cfg, batch, torch.device('cuda:0'))
    for _ in range(3):
        hp.train_step()
    torch.cuda.synchronize()
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            hp.train_step()
        torch.cuda.synchronize()
    by = collections.Counter()
    dev_us = collections.Counter()
    for ev in prof.events():
        if ev.device_time_total > 0 and (ev.name.startswith('aten::') or 'Memcpy' in ev.name or 'Memset' in ev.name):
            frames = [f for f in (ev.stack or []) if '/root/repo' in f or 'single_shot_detection_amd' in f or 'bench.py' in f]
            where = ' <- '.join(f.split('/')[-1] for f in frames[:3]) if frames else ('(no Python frame: autograd engine)' if not ev.stack else ev.stack[0].split('/')[-1])
            by[(ev.name, where)] += 1
            dev_us[(ev.name, where)] += ev.device_time_total
    rt = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith('hip') and ('Memcpy' in ev.name or 'Memset' in ev.name))
    print('runtime copies / memsets per step:', {k: v / steps for k, v in rt.items()})
    print(f'{cfg} batch {batch}: torch-side device operations per training step (calls, device us)')
    for (name, where), n in sorted(by.items(), key=lambda kv: -dev_us[kv[0]]):
        print(f'  {n / steps:6.1f}  {dev_us[(name, where)] / steps:8.1f} us  {name:22s} {where}')


if __name__ == '__main__':
    main()
