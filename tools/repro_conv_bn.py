"""One Conv2dBn case against torch CPU, with a report of WHERE the input gradient differs.  python3 tools/repro_conv_bn.py cin cout k stride pad hw B seed"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from single_shot_detection_amd.bf.modules import conv
from test_conv_bn_gpu import _RefConv2dBn, _randomize
cin, cout, k, stride, pad, hw, B, seed = [int(v) for v in sys.argv[1:9]]
rng = np.random.default_rng(seed)
m = conv.Conv2dBn(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)
_randomize(m, rng)
ref = _RefConv2dBn(m)
g = m.cuda()
x_np = rng.standard_normal((B, cin, hw, hw), dtype=np.float32)
for trial in range(3):
    xr = torch.from_numpy(x_np).requires_grad_(True)
    xg = torch.from_numpy(x_np).cuda().requires_grad_(True)
    g.train(); ref.train()
    yr, yg = ref(xr), g(xg)
    gy = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(yr.shape), dtype=np.float32))
    (yr * gy).sum().backward(); (yg * gy.cuda()).sum().backward()
    dy = np.abs(yg.detach().cpu().numpy() - yr.detach().numpy())
    dx = np.abs(xg.grad.cpu().numpy() - xr.grad.numpy())
    bad = np.argwhere(dx > 1e-3)
    print('trial', trial, 'y max diff %.2e' % dy.max(), 'dx max diff %.2e' % dx.max(), 'bad', len(bad))
    if len(bad):
        print('  bad b:', np.unique(bad[:, 0]), 'c range:', bad[:, 1].min(), bad[:, 1].max(), 'y:', np.unique(bad[:, 2])[:20], 'x:', np.unique(bad[:, 3])[:20])
    for p_ in list(g.parameters()) + list(ref.parameters()):
        p_.grad = None
