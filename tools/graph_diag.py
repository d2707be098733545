"""Where does a HIP-graph replay of the training step leave the eagerly enqueued step?  Two HotPaths from the same seed; one is warmed up on
a side stream and captured (like graphs.GraphedCallable), the other steps eagerly; prints the loss of every step of both and, after the
first replay, the parameters' gradients that differ most (the layer where the replay goes wrong).
    python3 tools/graph_diag.py <config> <batch> [other_stream] [no_defer] [no_prepare]
Environment knobs honoured by the library: SSDK_NO_FUSED_STATS, SSDK_NO_BN_CHAIN."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import ops  # noqa: E402
from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth  # noqa: E402


def named_params(hp):
    out = []
    for tag, m in (('heads', hp.heads), ('extras', hp.extras), ('tower', hp.tower), ('neck', hp.neck)):
        if m is not None:
            out += [(f'{tag}.{n}', p) for n, p in m.named_parameters() if not n.startswith('base.')]
    return out


def main():
    cfg, batch = sys.argv[1], int(sys.argv[2])
    flags = set(sys.argv[3:])
    dev = torch.device('cuda:0')
    eager, graphed = bench.HotPath(cfg, batch, dev), bench.HotPath(cfg, batch, dev)
    if 'no_defer' in flags:
        ops.defer_weight_gradients(False)
    if 'no_prepare' in flags:
        ops.prepare_weight_transposes = lambda module: 0
    graphed.gt = PackedGroundTruth.from_list(graphed.gt, dev, capacity=sum(len(g) for g in graphed.gt) + 7)
    le = [float(eager.train_step().detach()) for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lg = [float(graphed.train_step().detach()) for _ in range(3)]
    torch.cuda.current_stream().wait_stream(side)
    print('warm-up losses eager  ', le)
    print('warm-up losses graphed', lg, '(eager, on the side stream)')
    graph = torch.cuda.CUDAGraph()
    if 'other_stream' in flags:
        with torch.cuda.graph(graph):
            loss_g = graphed.train_step()
    else:
        with torch.cuda.graph(graph, stream=side):
            loss_g = graphed.train_step()
    for k in range(3):
        loss_e = eager.train_step()
        graph.replay()
        torch.cuda.synchronize()
        print('step %d: eager %.6f graph %.6f' % (k, float(loss_e.detach()), float(loss_g.detach())))
        if k == 0:
            rows = []
            for (n, p), (_, q) in zip(named_params(graphed), named_params(eager)):
                if p.grad is None or q.grad is None:
                    rows.append((float('inf'), n, 'grad missing: graph %s eager %s' % (p.grad is None, q.grad is None)))
                    continue
                d = float((p.grad - q.grad).abs().max())
                s = float(q.grad.abs().max()) + 1e-20
                rows.append((d / s, n, 'max|dg| %.3e of %.3e' % (d, s)))
            rows.sort(reverse=True)
            for r in rows[:12]:
                print('   %-60s rel %.3e  %s' % (r[1], r[0], r[2]))
            print('   ... median rel %.3e over %d parameters' % (sorted(r[0] for r in rows)[len(rows) // 2], len(rows)))


if __name__ == '__main__':
    main()
