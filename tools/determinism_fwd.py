"""First module of the neck / pyramid tail whose FORWARD output differs between runs on the same inputs (train mode; optionally with the
backward pass run in between, as in training).  python3 tools/determinism_fwd.py <config> <batch> [with_backward]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg, batch = sys.argv[1], int(sys.argv[2])
    with_bwd = 'with_backward' in sys.argv[3:]
    dev = torch.device('cuda:0')
    hp = bench.HotPath(cfg, batch, dev)
    roots = [(t, m) for t, m in (('neck', hp.neck), ('extras', hp.extras), ('tower', hp.tower)) if m is not None]
    runs = []
    for r in range(3):
        rec = []
        hooks = []
        for tag, root in roots:
            for name, m in root.named_modules():
                if name and not name.startswith('base') and len(list(m.children())) == 0 or type(m).__name__ in ('Conv2dBn', 'ThinnedUshapeModule'):
                    def hook(mod, inp, out, name=f'{tag}.{name}'):
                        outs = out if isinstance(out, (list, tuple)) else [out]
                        for i, o in enumerate(outs):
                            if isinstance(o, torch.Tensor):
                                rec.append((f'{name}[{i}]', o.detach().clone()))
                    hooks.append(m.register_forward_hook(hook))
        hp.opt.zero_grad(set_to_none=True)
        scores, locs = hp.forward_heads()
        rec.append(('scores', scores.detach().clone()))
        if with_bwd:
            target = hp.assigner.encode_ground_truth(hp.gt, hp.anchors)
            loss, _, _ = hp.criterion((scores, locs), hp.anchors, target)
            loss.backward()
        torch.cuda.synchronize()
        for h in hooks:
            h.remove()
        runs.append(rec)
    bad = 0
    for i, (name, t0) in enumerate(runs[0]):
        d = max(float((runs[r][i][1] - t0).abs().max()) for r in (1, 2))
        s = float(t0.abs().max()) + 1e-30
        n_bad = max(int(((runs[r][i][1] - t0).abs() > 1e-5 * s).sum()) for r in (1, 2))
        if d / s > 1e-5:
            bad += 1
            if bad <= 25:
                print('%-50s %-22s rel %.3e  elements off by > 1e-5: %d of %d' % (name, tuple(t0.shape), d / s, n_bad, t0.numel()))
    print('%d of %d recorded outputs differ by more than 1e-5 relative' % (bad, len(runs[0])))


if __name__ == '__main__':
    main()
