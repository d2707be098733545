"""Child process of tests/test_end_to_end_gpu.py::test_split_heads_step_captures_into_a_hip_graph: the sequence that died in round 4
(gpurun_out/r04a/probe1.log: Segmentation fault in torch/cuda/graphs.py capture_end <- graphs.py:42 <- tools/overlap_probe.py:70): the
training step with the heads split by dependency (multi_level_heads_split + side stream), two eager steps, capture into a HIP graph, replay.
Cause then: a reference cycle (join node -> shared object -> output tensor -> join node) kept every step's autograd graph -- and its
default-stream AccumulateGrad nodes -- alive.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import _lib  # noqa: E402
from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth  # noqa: E402
from single_shot_detection_amd.graphs import GraphedCallable  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    hp = bench.HotPath('ssd_300_vgg16_voc', 4, dev)
    hp.overlap = True                   # heads as two autograd nodes, the pyramid tail and its levels' heads on a side stream
    hp.one_launch = True
    hp.ordered_backward = True
    ref = bench.HotPath('ssd_300_vgg16_voc', 4, dev)   # the single-stream step from the same state
    ref.overlap = False
    with torch.no_grad():
        for p, q in zip(ref.params, hp.params):
            p.copy_(q)
    hp.gt = PackedGroundTruth.from_list(hp.gt, dev, capacity=sum(len(g) for g in hp.gt) + 7)
    ref.gt = hp.gt
    eager = [float(hp.train_step()) for _ in range(2)]           # two eager steps: their autograd graphs must be gone before the capture
    ref_losses = [float(ref.train_step()) for _ in range(5)]
    g = GraphedCallable(hp.train_step, [], warmup=2)             # two warm-up steps on the capture stream, then capture_begin ... capture_end (where round 4 crashed)
    g()
    replay = float(g.static_out.detach())
    torch.cuda.synchronize()
    print(json.dumps({'eager': eager, 'ref': ref_losses, 'replay': replay, 'timeouts': _lib.streamk_timeouts()}))


if __name__ == '__main__':
    main()
