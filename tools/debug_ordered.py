"""Debug probe of the ordered anchor-row backward: one level, a gradient on chosen anchors, every output against torch's CPU convolution."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from single_shot_detection_amd.detection.modules.heads import multi_level_heads
from test_heads_gpu import build_heads


def run(cin, h, nb, C, B, marks, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, h, h), dtype=np.float32)
    ws = rng.standard_normal((nb * C, cin, 3, 3), dtype=np.float32) * 0.05
    wl = rng.standard_normal((nb * 4, cin, 3, 3), dtype=np.float32) * 0.05
    bs = np.zeros(nb * C, np.float32)
    bl = np.zeros(nb * 4, np.float32)
    gs = np.zeros((B, h * h, nb, C), np.float32)
    gl = np.zeros((B, h * h, nb, 4), np.float32)
    for (b, p, k) in marks(B, h * h, nb, rng):
        gs[b, p, k] = rng.standard_normal(C)
        gl[b, p, k] = rng.standard_normal(4)
    xc = torch.from_numpy(x).requires_grad_(True)
    wsc, wlc = torch.from_numpy(ws).requires_grad_(True), torch.from_numpy(wl).requires_grad_(True)
    s = F.conv2d(xc, wsc, None, padding=1).permute(0, 2, 3, 1).reshape(B, -1)
    l = F.conv2d(xc, wlc, None, padding=1).permute(0, 2, 3, 1).reshape(B, -1)
    ((s * torch.from_numpy(gs).view(B, -1)).sum() + (l * torch.from_numpy(gl).view(B, -1)).sum()).backward()
    heads = build_heads([(cin, h, nb)], C, {('score', 0): (ws, bs), ('loc', 0): (wl, bl)})
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    sg, lg = multi_level_heads([xg], [xg], heads)
    ((sg * torch.from_numpy(gs).view(B, -1).cuda()).sum() + (lg * torch.from_numpy(gl).view(B, -1).cuda()).sum()).backward()
    dx, dxr = xg.grad.cpu().numpy(), xc.grad.numpy()
    err = np.abs(dx - dxr)
    print(f'cin={cin} h={h} nb={nb} C={C} B={B}: dx max err {err.max():.3e} (scale {np.abs(dxr).max():.3e})', end='  ')
    for name, got, ref in (('dws', heads[0]['score'].weight.grad.cpu().numpy(), wsc.grad.numpy()), ('dwl', heads[0]['loc'].weight.grad.cpu().numpy(), wlc.grad.numpy()),
                           ('dbs', heads[0]['score'].bias.grad.cpu().numpy(), gs.sum((0, 1)).reshape(-1)), ('dbl', heads[0]['loc'].bias.grad.cpu().numpy(), gl.sum((0, 1)).reshape(-1))):
        print(f'{name} {np.abs(got - ref).max():.3e}/{np.abs(ref).max():.2e}', end='  ')
    print()
    if err.max() > 1e-3 * np.abs(dxr).max():
        bad = np.argwhere(err > 1e-3 * np.abs(dxr).max())
        print('   bad dx elements', len(bad), 'of', err.size, 'first', bad[:5].tolist(), 'channels hit', np.unique(bad[:, 1])[:16].tolist(), 'pixels', np.unique(bad[:, 2] * h + bad[:, 3])[:16].tolist())
        wbad = np.abs(heads[0]['score'].weight.grad.cpu().numpy() - wsc.grad.numpy())
        print('   dws bad rows', np.unique(np.argwhere(wbad > 1e-3 * wbad.max() + 1e-6)[:, 0])[:20].tolist())


one = lambda B, P, nb, rng: [(0, P // 2, 0)]
every = lambda B, P, nb, rng: [(b, p, k) for b in range(B) for p in range(P) for k in range(nb)]
some = lambda B, P, nb, rng: [(b, p, k) for b in range(B) for p in range(P) for k in range(nb) if rng.random() < 0.1]
for cfg in ((32, 4, 2, 3, 1), (32, 4, 2, 4, 1), (32, 4, 2, 5, 1), (32, 4, 2, 8, 1), (32, 4, 2, 9, 1), (32, 4, 2, 12, 1), (32, 4, 2, 13, 1), (32, 4, 2, 28, 1)):
    for marks in (one,):
        run(*cfg, marks)
