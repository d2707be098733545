"""Experiment: the whole training step (pyramid tail + heads forward, match, sampler, loss, backward, fused SGD) captured in a HIP graph.
The ground truth is packed once into static device buffers (the per-step host-side packing + H2D copy cannot be captured).
    python3 tools/graph_train.py [config] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd.detection import target_assigner as ta  # noqa: E402


def timeit(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_mb2_voc'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device('cuda:0')
    hp = bench.HotPath(cfg, batch, dev)
    for _ in range(3):
        hp.train_step()
    eager = timeit(hp.train_step, 30)
    packed = ta.pack_ground_truth(hp.gt, dev)
    orig = ta.pack_ground_truth
    ta.pack_ground_truth = lambda gt, device, row=6: packed
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                hp.train_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = hp.train_step()
        graph.replay()
        torch.cuda.synchronize()
        l0 = float(loss.detach())
        replay = timeit(graph.replay, 30)
        l1 = float(loss.detach())
    finally:
        ta.pack_ground_truth = orig
    print('%s b%d: eager %.3f ms/step, graph replay %.3f ms/step; loss %.5f -> %.5f after 30 more replays (training goes on)'
          % (cfg, batch, eager * 1e3, replay * 1e3, l0, l1))
    # the same number of steps taken eagerly from the same initial state must arrive at the same loss (up to atomics' rounding)
    hp2 = bench.HotPath(cfg, batch, dev)
    steps = 3 + 30 + 3 + 1 + 30   # every step hp EXECUTED before l1 was read: warm-up, timing, side-stream warm-up, one replay, 30 timed replays
    # (the capture itself executes nothing; round 2 counted 74 here and compared the 67th graph step with the 74th eager one)
    for _ in range(steps - 1):
        hp2.train_step()
    print('eager loss after the same %d steps: %.5f' % (steps, float(hp2.train_step().detach())))


if __name__ == '__main__':
    main()
