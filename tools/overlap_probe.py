#!/usr/bin/env python3
"""The training step with the heads split by dependency (backbone-tap levels beside the pyramid tail, bench.HotPath.overlap) against the
single-stream step: ms per step eager and replayed from a HIP graph, for several caps on the main GEMM's persistent workgroups.
    python tools/overlap_probe.py [config] [batch]
"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import _lib  # noqa: E402
from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth  # noqa: E402
from single_shot_detection_amd.graphs import GraphedCallable  # noqa: E402


def ms(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    gc.enable()
    return dt


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device('cuda:0')
    variants = [('single stream (round 3)', dict(overlap=False)),
                ('split bwd, one stream', dict(overlap=True, one_stream=True, one_launch=True)),
                ('split bwd, side stream', dict(overlap=True, one_launch=True)),
                ('split bwd, side stream, ordered', dict(overlap=True, one_launch=True, ordered=True)),
                ('split fwd+bwd, side stream, main 512', dict(overlap=True, main=512, side=256)),
                ('split fwd+bwd, side stream, main 480', dict(overlap=True, main=480, side=256)),
                ('split fwd+bwd, side stream, main 448', dict(overlap=True, main=448, side=256)),
                ('split fwd+bwd, side, main 480, ordered', dict(overlap=True, main=480, side=256, ordered=True))]
    ref_loss = None
    only = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else range(len(variants))
    for name, v in [variants[i] for i in only]:
        hp = bench.HotPath(cfg, batch, dev)
        hp.overlap = v['overlap'] and hp.extras is not None
        hp.main_workgroups = v.get('main', 0)
        hp.side_workgroups = v.get('side', 0)
        hp.one_launch = v.get('one_launch', False)
        hp.ordered_backward = v.get('ordered', False)
        if v.get('one_stream'):
            os.environ['SSDK_OVERLAP_ONE_STREAM'] = '1'
        else:
            os.environ.pop('SSDK_OVERLAP_ONE_STREAM', None)
        loss0 = float(hp.train_step())
        if ref_loss is None:
            ref_loss = loss0
        eager = ms(hp.train_step)
        hp.fwd_events, hp.fwd_steps = [], 0
        for _ in range(10):
            hp.train_step(timed=True)
        torch.cuda.synchronize()
        gemm_ms, fl = bench.head_gemm_time(hp)
        hp.gt = PackedGroundTruth.from_list(hp.gt, dev, capacity=sum(len(g) for g in hp.gt) + 7)
        try:
            g = GraphedCallable(hp.train_step, [], warmup=2)
            replay = ms(g)
            lossg = float(g.static_out.detach())
        except Exception as e:
            replay, lossg = float('nan'), repr(e)[:200]
        print(f'{name:42s} eager {eager:6.3f} ms  replay {replay:6.3f} ms  main GEMM {gemm_ms:6.3f} ms = {fl * batch / gemm_ms / 1e9 / bench.PEAK_FP32_MATRIX_TFLOPS:5.3f} of peak'
              f'  first loss {loss0:.6f} (ref {ref_loss:.6f})  loss after replays {lossg}  timeouts {_lib.streamk_timeouts()}', flush=True)
        del hp
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
