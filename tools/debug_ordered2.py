"""Debug probe 2: the intermediates of the ordered anchor-row backward (ga, aidx, T) against numpy, through ssdk_debug_heads_bwd_layout."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from single_shot_detection_amd import _lib
from single_shot_detection_amd.detection.modules import heads as H
from single_shot_detection_amd.detection.modules.heads import multi_level_heads
from test_heads_gpu import build_heads


def run(cin, h, nb, Cc, B, marks, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, h, h), dtype=np.float32)
    ws = rng.standard_normal((nb * Cc, cin, 3, 3), dtype=np.float32) * 0.05
    wl = rng.standard_normal((nb * 4, cin, 3, 3), dtype=np.float32) * 0.05
    gs = np.zeros((B, h * h, nb, Cc), np.float32)
    gl = np.zeros((B, h * h, nb, 4), np.float32)
    for (b, p, k) in marks(B, h * h, nb, rng):
        gs[b, p, k] = rng.standard_normal(Cc)
        gl[b, p, k] = rng.standard_normal(4)
    heads = build_heads([(cin, h, nb)], Cc, {('score', 0): (ws, np.zeros(nb * Cc, np.float32)), ('loc', 0): (wl, np.zeros(nb * 4, np.float32))})
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    sg, lg = multi_level_heads([xg], [xg], heads)
    ((sg * torch.from_numpy(gs).view(B, -1).cuda()).sum() + (lg * torch.from_numpy(gl).view(B, -1).cuda()).sum()).backward()
    torch.cuda.synchronize()
    lv = dict(x=H.to_nhwc(xg.detach()), H=h, W=h, cin=cin, ws=H.weight_khwc(heads[0]['score'].weight.detach()), bs=None, wl=H.weight_khwc(heads[0]['loc'].weight.detach()),
              bl=None, ns=nb * Cc, nl=nb * 4, s_off=0, l_off=0)
    arr = H._level_array([lv])
    out = (C.c_ulonglong * 8)()
    rc = _lib.lib().ssdk_debug_heads_bwd_layout(arr, 1, B, 0, out)
    assert rc == 0, rc
    wsb = _lib.scratch(0, xg.device, 'heads_bwd').cpu().numpy()
    M, K9, Jpad = B * h * h, 9 * cin, (Cc + 4 + 31) // 32 * 32
    i32 = lambda off, n: wsb[off:off + 4 * n].view(np.int32)
    f32 = lambda off, n: wsb[off:off + 4 * n].view(np.float32)
    acounts, plan, mode, tcap = i32(out[4], 16), i32(out[5], 34), int(i32(out[6], 1)[0]), int(out[7])
    print(f'cin={cin} h={h} nb={nb} C={Cc} B={B}: mode {mode} counts {acounts[:nb].tolist()} tbase {plan[:nb + 1].tolist()} tiles {plan[17:17 + nb + 1].tolist()} tcap {tcap}')
    ga = f32(out[0], nb * M * Jpad).reshape(nb, M, Jpad)
    apix = i32(out[3], nb * M).reshape(nb, M)
    aidx = i32(out[2], nb * M).reshape(nb, M)
    rows = int(plan[nb])
    T = f32(out[1], rows * K9).reshape(rows, K9)
    wsk = ws.transpose(0, 2, 3, 1).reshape(nb, Cc, K9)     # [k][j][tap * cin + c]
    wlk = wl.transpose(0, 2, 3, 1).reshape(nb, 4, K9)
    for k in range(nb):
        n = int(acounts[k])
        if not n:
            continue
        pix = apix[k, :n]
        assert np.all(np.diff(pix) > 0), 'rows not in pixel order'
        b_, p_ = pix // (h * h), pix % (h * h)
        ref_rows = np.concatenate([gs[b_, p_, k], gl[b_, p_, k]], 1)
        e_ga = np.abs(ga[k, :n, :Cc + 4] - ref_rows).max()
        assert np.all(ga[k, :n, Cc + 4:] == 0)
        assert np.array_equal(aidx[k, pix], plan[k] + np.arange(n)), 'aidx'
        assert (aidx[k] >= 0).sum() == n
        Wk = np.concatenate([wsk[k], wlk[k]], 0)            # [C + 4][K9]
        Tref = ref_rows.astype(np.float64) @ Wk.astype(np.float64)
        Tk = T[plan[k]:plan[k] + n]
        err = np.abs(Tk - Tref)
        print(f'   type {k}: rows {n} ga err {e_ga:.2e}  T err {err.max():.3e} (scale {np.abs(Tref).max():.2e})')
        if err.max() > 1e-3:
            # which j are missing?  least squares of the residual against the rows of Wk
            res = (Tk - Tref)[0]
            coef, *_ = np.linalg.lstsq(Wk.T.astype(np.float64), res.astype(np.float64), rcond=None)
            print('      residual of row 0 in terms of W rows (coef / the row\'s own gradient value):', np.round(coef / (ref_rows[0] + 1e-30), 2).tolist())
            badcols = np.argwhere(err[0] > 1e-3).ravel()
            print('      bad columns of row 0:', len(badcols), 'of', K9, badcols[:12].tolist())


one = lambda B, P, nb, rng: [(0, P // 2, 0)]
some = lambda B, P, nb, rng: [(b, p, k) for b in range(B) for p in range(P) for k in range(nb) if rng.random() < 0.1]
for cfg in ((32, 4, 2, 4, 1), (32, 4, 2, 5, 1), (32, 4, 2, 12, 1), (32, 4, 2, 28, 1)):
    run(*cfg, one)
run(64, 6, 4, 21, 2, some)
every = lambda B, P, nb, rng: [(b, p, k) for b in range(B) for p in range(P) for k in range(nb)]
run(512, 18, 6, 81, 2, every)
