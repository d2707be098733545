"""Is one training step a function of its state?  Runs forward + backward (no optimizer step) several times from the SAME parameters and
inputs and prints, per parameter, the largest relative difference of its gradient between runs (fp32 atomics reorder sums: ~1e-6 is
expected; anything near 1 is a race or an uninitialised read).  python3 tools/determinism_diag.py <config> <batch> [runs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tools.graph_diag import named_params  # noqa: E402


def main():
    cfg, batch = sys.argv[1], int(sys.argv[2])
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device('cuda:0')
    hp = bench.HotPath(cfg, batch, dev)
    if os.environ.get('DIAG_NO_DEFER'):
        from single_shot_detection_amd import ops
        ops.defer_weight_gradients(False)
    grads, losses, in_grads = [], [], []
    for r in range(runs):
        hp.opt.zero_grad(set_to_none=True)
        for s in hp.inputs:
            s.grad = None
        scores, locs = hp.forward_heads()
        target = hp.assigner.encode_ground_truth(hp.gt, hp.anchors)
        loss, _, _ = hp.criterion((scores, locs), hp.anchors, target)
        loss.backward()
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
        grads.append({n: p.grad.detach().clone() for n, p in named_params(hp)})
        in_grads.append([s.grad.detach().clone() for s in hp.inputs])
    print('losses', losses)
    rows = []
    for n in grads[0]:
        d = max(float((grads[r][n] - grads[0][n]).abs().max()) for r in range(1, runs))
        s = float(grads[0][n].abs().max()) + 1e-30
        rows.append((d / s, n, d, s))
    order = [n for n, _ in named_params(hp)]
    rows.sort(reverse=True)
    for rel, n, d, s in rows[:25]:
        print('  %-58s rel %.3e (max|dg| %.3e of %.3e) #%d' % (n, rel, d, s, order.index(n)))
    import collections, re
    grp = collections.defaultdict(list)
    for rel, n, d, s_ in rows:
        key = re.sub(r'\.(conv|bn)\.(weight|bias)$', '', n)
        key = re.sub(r'\.\d+$', '', key)
        grp[key].append(rel)
    for k in sorted(grp):
        v = grp[k]
        print('    group %-40s n %3d  max rel %.2e  min rel %.2e' % (k, len(v), max(v), min(v)))
    print('  median rel %.3e, parameters above 1e-3: %d of %d' % (sorted(r[0] for r in rows)[len(rows) // 2], sum(r[0] > 1e-3 for r in rows), len(rows)))
    for i in range(len(in_grads[0])):
        d = max(float((in_grads[r][i] - in_grads[0][i]).abs().max()) for r in range(1, runs))
        print('  input %d grad: max diff %.3e of %.3e' % (i, d, float(in_grads[0][i].abs().max())))


if __name__ == '__main__':
    main()
