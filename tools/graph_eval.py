"""Experiment: the evaluation step (pyramid tail + heads forward + postprocess) captured in a HIP graph (torch.cuda.CUDAGraph) and
replayed, against the eagerly enqueued step.   python3 tools/graph_eval.py [config] [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timeit(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'ssd_300_vgg16_voc'
    batches = [int(b) for b in sys.argv[2:]] or [1, 2, 8, 32]
    dev = torch.device('cuda:0')
    for batch in batches:
        hp = bench.HotPath(cfg, batch, dev)
        for _ in range(3):
            ref = hp.eval_step()
        eager = timeit(hp.eval_step, 30)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                hp.eval_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = hp.eval_step()
        graph.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, out) if isinstance(a, torch.Tensor))
        replay = timeit(graph.replay, 30)
        print('%s b%d: eager %.1f us (%.0f img/s), graph replay %.1f us (%.0f img/s), outputs equal: %s'
              % (cfg, batch, eager * 1e6, batch / eager, replay * 1e6, batch / replay, same))


if __name__ == '__main__':
    main()
