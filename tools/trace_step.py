"""Which torch ops (not libssdk launches) does one train step of bench.HotPath enqueue, and from where?
python3 tools/trace_step.py [config] [batch]   -> table of aten ops that launched a kernel or a copy, with the Python frames."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device('cuda:0')
    hp = bench.HotPath(cfg, batch, dev)
    for _ in range(3):
        hp.train_step()
    torch.cuda.synchronize()
    steps = 4
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            hp.train_step()
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_stack_n=6):
        dev_us = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
        self_dev = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
        if self_dev <= 0 or not ev.key.startswith('aten::'):
            continue
        stack = [f for f in ev.stack if 'site-packages' not in f and 'torch/' not in f][:3]
        rows.append((self_dev / steps, ev.count / steps, ev.key, ' <- '.join(s.strip() for s in stack)))
    rows.sort(reverse=True)
    print('self device us/step | calls/step | op | frames')
    for us, n, key, st in rows:
        print('%8.1f | %5.1f | %-28s | %s' % (us, n, key, st))
    kern = [(e.self_device_time_total / steps, e.count / steps, e.key) for e in prof.key_averages()
            if getattr(e, 'device_type', None) is not None and e.self_device_time_total > 0 and not e.key.startswith('aten::')]
    kern.sort(reverse=True)
    print('\nkernels: us/step | launches/step | name')
    tot_n = 0
    for us, n, key in kern:
        tot_n += n
        print('%8.1f | %5.1f | %s' % (us, n, key[:110]))
    print('launches per step: %.1f' % tot_n)


if __name__ == '__main__':
    main()
