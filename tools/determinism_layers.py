"""Which layer shape of a config's neck / pyramid tail is not a function of its inputs?  Records the input shape of every Conv2dBn of the
model in one forward pass, then runs each distinct (cin, cout, k, stride, pad, H, W) block alone, forward + backward, three times on the same
input and prints the largest relative difference between runs of y, dx and every parameter gradient.  python3 tools/determinism_layers.py <config> <batch>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import ops  # noqa: E402
from single_shot_detection_amd.bf.modules import conv  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def main():
    cfg, batch = sys.argv[1], int(sys.argv[2])
    dev = torch.device('cuda:0')
    hp = bench.HotPath(cfg, batch, dev)
    if os.environ.get('DIAG_NO_DEFER'):
        ops.defer_weight_gradients(False)
    shapes = {}
    hooks = []
    for root in (hp.neck, hp.extras, hp.tower):
        if root is None:
            continue
        for name, m in root.named_modules():
            if isinstance(m, conv.Conv2dBn):
                def hook(mod, inp, out, name=name):
                    c = mod.conv
                    key = (c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], tuple(inp[0].shape[2:]), 'bn' in mod._modules, 'activation' in mod._modules)
                    shapes.setdefault(key, (name, mod))
                hooks.append(m.register_forward_hook(hook))
    hp.train_step()
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    print(len(shapes), 'distinct Conv2dBn shapes')
    rng = np.random.default_rng(3)
    for key, (name, mod) in sorted(shapes.items(), key=lambda kv: kv[1][0]):
        cin, cout, k, s, p, hw, has_bn, has_act = key
        x_np = rng.standard_normal((batch, cin) + hw, dtype=np.float32)
        res = []
        gy = None
        for r in range(3):
            mod.zero_grad(set_to_none=True)
            x = torch.from_numpy(x_np).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = mod(x)
            if gy is None:
                gy = torch.from_numpy(rng.standard_normal(tuple(y.shape), dtype=np.float32)).to(dev).contiguous(memory_format=torch.channels_last)
            y.backward(gy)
            torch.cuda.synchronize()
            res.append([y.detach().clone(), x.grad.detach().clone()] + [q.grad.detach().clone() for q in mod.parameters()])
        worst = [max(rel(res[r][i], res[0][i]) for r in (1, 2)) for i in range(len(res[0]))]
        flag = '  <-- NOT DETERMINISTIC' if max(worst) > 1e-4 else ''
        print('%-40s cin %4d cout %4d k%d s%d p%d %-10s y %.1e dx %.1e dparams %s%s' % (name, cin, cout, k, s, p, hw, worst[0], worst[1],
                                                                                         ' '.join('%.1e' % w for w in worst[2:]), flag), flush=True)
    # the neck's other ops
    if hp.neck is not None:
        for C, hf, hc in ((256, 64, 32), (256, 4, 2), (768, 64, 32)):
            fine_np, coarse_np = rng.standard_normal((batch, C, hf, hf), dtype=np.float32), rng.standard_normal((batch, C, hc, hc), dtype=np.float32)
            outs = []
            for r in range(3):
                f = torch.from_numpy(fine_np).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                c = torch.from_numpy(coarse_np).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                o = ops.upsample_add(f, c)
                (o * o).sum().backward()
                outs.append((o.detach().clone(), f.grad.clone(), c.grad.clone()))
            print('upsample_add C %d %d<-%d:' % (C, hf, hc), ['%.1e' % max(rel(outs[r][i], outs[0][i]) for r in (1, 2)) for i in range(3)])


if __name__ == '__main__':
    main()
