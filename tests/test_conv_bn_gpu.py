"""GPU parity: generic NHWC conv + BatchNorm/ReLU (H2 extras, H3 tower) through the C ABI vs torch CPU modules
(fp32 reference of the same op), forward and backward, training and eval BatchNorm modes."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from single_shot_detection_amd.bf.modules import conv
from single_shot_detection_amd.detection import detector_builder
from single_shot_detection_amd.detection.modules import predictors

pytestmark = pytest.mark.gpu


def _close(got, want, bar=1e-4, err_msg='', scale=None):
    """north_star's fp32 bar: |got - want| <= 1e-4 * (|want| + max|want|) for every element -- relative to the tensor's own scale, because the
    GPU's GEMMs and reductions sum in another order than torch's CPU kernels and an element that is a near-cancellation of K products
    cannot be held to a fraction of itself.  (The same compositions against REFERENCE-generated fixtures: test_blocks_golden_gpu.py, 2e-5.)
    ``scale``: the magnitude to use instead of max|want| -- for a gradient that is analytically ZERO (a bias in front of a BatchNorm: the
    norm subtracts the mean) both sides hold nothing but rounding noise of the sums they came from, and the scale is that of the sibling
    weight gradient."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (err_msg, got.shape, want.shape)
    if scale is None:
        scale = float(np.abs(want).max()) if want.size else 0.0
    err = np.abs(got - want)
    tol = bar * (np.abs(want) + scale) + 1e-12
    bad = err > tol
    assert not bad.any(), (err_msg, int(bad.sum()), float((err / tol).max()), scale)


class _RefConv2dBn(nn.Module):
    """bf/modules/conv.py:4-36 with stock torch ops (CPU reference)."""
    def __init__(self, m):
        super().__init__()
        self.conv = copy.deepcopy(m.conv)
        self.bn = copy.deepcopy(m.bn) if 'bn' in m._modules else None
        self.act = 'activation' in m._modules

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return torch.relu(x) if self.act else x


def _randomize(m, rng):
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape), dtype=np.float32) * (0.1 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 else 0.0)))


def _compare_module(gpu_m, ref_m, x_np, train, rtol=2e-4, atol=2e-4):
    gpu_m.train(train); ref_m.train(train)
    xr = torch.from_numpy(x_np).requires_grad_(True)
    xg = torch.from_numpy(x_np).cuda().requires_grad_(True)
    yr = ref_m(xr)
    yg = gpu_m(xg)
    _close(yg.detach().cpu().numpy(), yr.detach().numpy())
    g = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(yr.shape), dtype=np.float32))
    (yr * g).sum().backward()
    (yg * g.cuda()).sum().backward()
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())
    for (n1, p1), (n2, p2) in zip(sorted(gpu_m.named_parameters()), sorted(ref_m.named_parameters())):
        if p2.grad is None:
            assert p1.grad is None or not p1.requires_grad
            continue
        scale = float(p2.grad.abs().max()) + 1e-6
        _close(p1.grad.cpu().numpy(), p2.grad.numpy(), err_msg=n1)
    for (n1, b1), (n2, b2) in zip(sorted(gpu_m.named_buffers()), sorted(ref_m.named_buffers())):
        np.testing.assert_allclose(b1.cpu().numpy(), b2.numpy(), rtol=1e-4, atol=1e-5, err_msg=n1)


class _RefDepthwise(nn.Module):
    """bf/modules/conv.py:39-85 with stock torch ops (CPU reference)."""
    def __init__(self, m):
        super().__init__()
        self.mods = nn.ModuleDict({k: copy.deepcopy(v) for k, v in m._modules.items()})

    def forward(self, x):
        for name in ('depthwise_conv', 'depthwise_bn', 'depthwise_activation', 'pointwise_conv', 'pointwise_bn', 'pointwise_activation'):
            if name in self.mods:
                x = self.mods[name](x)
        return x


def _ref_of(m):
    return _RefDepthwise(m) if isinstance(m, conv.DepthwiseConv2dBn) else _RefConv2dBn(m)


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('cin,cout,k,stride,pad,hw', [(512, 256, 1, 1, 0, 18), (256, 512, 3, 2, 1, 18), (128, 256, 3, 2, 1, 5),
                                                      (128, 256, 3, 1, 0, 3), (64, 32, 3, 2, 1, 2)])
def test_conv2dbn_block_vs_torch(cin, cout, k, stride, pad, hw, train):
    rng = np.random.default_rng(7)
    m = conv.Conv2dBn(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)
    _randomize(m, rng)
    with torch.no_grad():
        m.bn.running_mean.copy_(torch.from_numpy(rng.standard_normal(cout, dtype=np.float32) * 0.1))
        m.bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 2.0, cout).astype(np.float32)))
    ref = _RefConv2dBn(m)
    gpu = m.cuda()
    x = rng.standard_normal((4, cin, hw, hw), dtype=np.float32)
    _compare_module(gpu, ref, x, train)


def test_ssd_extras_chain_vs_torch():
    rng = np.random.default_rng(11)
    extras = detector_builder.get_extras([512], layers=(('s', 512), ('s', 256), ('s', 256), ('s', 256)))   # ssd_300_vgg16_voc
    _randomize(extras, rng)
    ref = nn.ModuleList([nn.Sequential(*[_RefConv2dBn(b) for b in blk]) for blk in extras])
    extras = extras.cuda()
    x_np = rng.standard_normal((2, 512, 18, 18), dtype=np.float32)
    xr = torch.from_numpy(x_np)
    xg = torch.from_numpy(x_np).cuda()
    for blk_g, blk_r, want in zip(extras, ref, (9, 5, 3, 2)):
        xr = blk_r(xr)
        xg = blk_g(xg)
        assert xg.shape[2] == want
        _close(xg.detach().cpu().numpy(), xr.detach().numpy())


@pytest.mark.parametrize('train', [True, False])
def test_retina_tower_vs_torch(train):
    rng = np.random.default_rng(13)
    sizes = [16, 8, 4, 3, 2]   # >= 16 rows per BatchNorm: BN over 2 samples is ill-conditioned (rstd up to 1/sqrt(eps))
    tower = predictors.SharedConvPredictor([64] * 5, [9] * 5, 80, False, num_layers=2, num_channels=64)
    _randomize(tower, rng)
    ref = copy.deepcopy(tower)
    # reference forward with stock torch ops: conv -> ReLU -> per-level BN (predictors.py:60-76)
    def ref_forward(srcs):
        s = l = srcs
        for sc, lc, sn, ln in zip(ref.convs['score'], ref.convs['loc'], ref.norms['score'], ref.norms['loc']):
            s = [n(torch.relu(sc.conv(x))) for n, x in zip(sn, s)]
            l = [n(torch.relu(lc.conv(x))) for n, x in zip(ln, l)]
        return s, l
    tower = tower.cuda()
    tower.train(train); ref.train(train)
    xs_np = [rng.standard_normal((4, 64, h, h), dtype=np.float32) for h in sizes]
    xr = [torch.from_numpy(x).requires_grad_(True) for x in xs_np]
    xg = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs_np]
    sr, lr = ref_forward(xr)
    sg, lg = tower(xg)
    gw = [torch.from_numpy(rng.standard_normal(tuple(a.shape), dtype=np.float32)) for a in sr + lr]   # sum(BN(x)) alone has zero gradient
    tot_r = sum((a * a).sum() for a in sr) + sum((a * g).sum() for a, g in zip(sr + lr, gw))
    tot_g = sum((a * a).sum() for a in sg) + sum((a * g.cuda()).sum() for a, g in zip(sg + lg, gw))
    for a, b in zip(sg + lg, sr + lr):
        _close(a.detach().cpu().numpy(), b.detach().numpy())
    tot_r.backward(); tot_g.backward()
    # the 1x1 level normalises over 2 samples: dead channels carry rstd = 1/sqrt(eps) = 316 and amplify fp32 rounding
    # of the upstream gradients, so the tolerance is relative to the largest gradient of the tensor
    for a, b in zip(xg, xr):
        _close(a.grad.cpu().numpy(), b.grad.numpy())
    for (n1, p1), (n2, p2) in zip(sorted(tower.named_parameters()), sorted(ref.named_parameters())):
        scale = float(p2.grad.abs().max()) + 1e-6
        _close(p1.grad.cpu().numpy(), p2.grad.numpy(), err_msg=n1)


def test_fpn_neck_vs_torch():
    """FeaturePyramid (bf/modules/features.py:52-120) on libssdk vs the same graph on stock torch CPU ops."""
    import torch.nn.functional as F
    from single_shot_detection_amd.bf.modules.features import FeaturePyramid

    class _Base(nn.Module):   # three taps at strides 1, 2, 4 with odd sizes (31 -> 16 -> 8) like ResNet's 63 -> 32 -> 16
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.Conv2d(8, 16, 3, stride=2, padding=1), nn.Conv2d(16, 32, 3, stride=2, padding=1))

    rng = np.random.default_rng(5)
    torch.manual_seed(0)
    fpn = FeaturePyramid(_Base(), out_layers=(0, 1, 2), pyramid_layers=5, pyramid_channels=16)
    _randomize(fpn.pyramid_lateral, rng); _randomize(fpn.pyramid_output, rng)
    ref = copy.deepcopy(fpn)
    fpn = fpn.cuda()
    x_np = rng.standard_normal((2, 3, 31, 31), dtype=np.float32)

    def ref_forward(x):
        srcs, cur = [], x
        for layer in ref.base:
            cur = layer(cur); srcs.append(cur)
        feats = [lat(s) for s, lat in zip(srcs, ref.pyramid_lateral)]
        for i in reversed(range(len(feats) - 1)):
            feats[i] = feats[i] + F.interpolate(feats[i + 1], size=feats[i].shape[2:], mode='nearest')
        outs = []
        for k, blk in enumerate(ref.pyramid_output):
            src = feats[k] if k < len(feats) else outs[-1]
            outs.append(torch.relu(blk.bn(blk.conv(src))))
        return outs

    xr = torch.from_numpy(x_np).requires_grad_(True)
    xg = torch.from_numpy(x_np).cuda().requires_grad_(True)
    outs_r = ref_forward(xr)
    outs_g, last = fpn(xg)
    assert [tuple(o.shape[2:]) for o in outs_g] == [(31, 31), (16, 16), (8, 8), (4, 4), (2, 2)] and last is outs_g[-1]
    gws = [torch.from_numpy(rng.standard_normal(tuple(o.shape), dtype=np.float32)) for o in outs_r]
    for a, b in zip(outs_g, outs_r):
        _close(a.detach().cpu().numpy(), b.detach().numpy())
    sum((a * g).sum() for a, g in zip(outs_r, gws)).backward()
    sum((a * g.cuda()).sum() for a, g in zip(outs_g, gws)).backward()
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())
    for (n1, p1), (n2, p2) in zip(sorted(fpn.named_parameters()), sorted(ref.named_parameters())):
        scale = float(p2.grad.abs().max()) + 1e-6
        _close(p1.grad.cpu().numpy(), p2.grad.numpy(), err_msg=n1)


def test_m2det_neck_vs_torch():
    """MultilevelFeaturePyramid (TUM + SFAM, bf/modules/features.py:215-393) on libssdk vs the same graph on stock torch CPU ops."""
    import torch.nn.functional as F
    from single_shot_detection_amd.bf.modules.features import MultilevelFeaturePyramid

    class _Base(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.Conv2d(8, 16, 3, stride=2, padding=1))

    rng = np.random.default_rng(9)
    torch.manual_seed(0)
    neck = MultilevelFeaturePyramid(_Base(), out_layers=(0, 1), num_scales=3, num_tums=3, base_reduced_channels=[16, 8], reduced_channels=8,   # (three TUMs: two reducers, merged into one convolution)
                                    tum={'inner_channels': 16, 'out_channels': 8}, sfam={'reduction_ratio': 2})
    for grp in (neck.base_reducers, neck.tums, neck.reducers, neck.sfam):
        _randomize(grp, rng)
    ref = copy.deepcopy(neck)
    neck = neck.cuda()
    x_np = rng.standard_normal((4, 3, 21, 21), dtype=np.float32)

    def blk(b, x):   # Conv2dBn with stock ops
        return torch.relu(b.bn(b.conv(x)))

    def tum_ref(t, x):
        down = [x]
        for l in t.down_layers:
            x = blk(l, x); down.append(x)
        up = [x]
        for dx, l in zip(reversed(down[:-1]), reversed(t.up_layers)):
            x = F.interpolate(blk(l, x), size=dx.shape[2:], mode='nearest') + dx
            up.append(x)
        return [blk(l, u) for l, u in zip(reversed(t.smooth_layers), up)]

    def ref_forward(x):
        srcs, cur = [], x
        for layer in ref.base:
            cur = layer(cur); srcs.append(cur)
        br = [blk(r, s) for r, s in zip(ref.base_reducers, srcs)]
        base = torch.cat([br[0]] + [F.interpolate(f, size=br[0].shape[2:], mode='nearest') for f in br[1:]], dim=1)
        feats = [[f] for f in tum_ref(ref.tums[0], base)]
        for t, r in zip(ref.tums[1:], ref.reducers):
            xx = torch.cat([feats[-1][-1], blk(r, base)], dim=1)
            for i, f in enumerate(tum_ref(t, xx)):
                feats[i].append(f)
        feats = [torch.cat(f, dim=1) for f in reversed(feats)]
        out = []
        for f, fc1, fc2 in zip(feats, ref.sfam.fc1, ref.sfam.fc2):
            z = fc2(torch.relu(fc1(F.adaptive_avg_pool2d(f, 1))))
            out.append(f * torch.sigmoid(z))
        return out

    xr = torch.from_numpy(x_np).requires_grad_(True)
    xg = torch.from_numpy(x_np).cuda().requires_grad_(True)
    outs_r = ref_forward(xr)
    outs_g, _ = neck(xg)
    assert [tuple(o.shape) for o in outs_g] == [tuple(o.shape) for o in outs_r]
    for a, b in zip(outs_g, outs_r):
        _close(a.detach().cpu().numpy(), b.detach().numpy())
    gws = [torch.from_numpy(rng.standard_normal(tuple(o.shape), dtype=np.float32)) for o in outs_r]
    sum((a * g).sum() for a, g in zip(outs_r, gws)).backward()
    sum((a * g.cuda()).sum() for a, g in zip(outs_g, gws)).backward()
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())
    gw_max = max(float(p.grad.abs().max()) for p in ref.parameters())
    for (n1, p1), (n2, p2) in zip(sorted(neck.named_parameters()), sorted(ref.named_parameters())):
        # biases feeding a BatchNorm (the stub taps' convolutions sit in front of the reducers' norms only through a ReLU-free path when
        # their gradient is analytically zero): what is left is rounding noise of sums of the weight gradient's magnitude
        zero_grad_bias = n1.endswith('.bias') and float(p2.grad.abs().max()) < 1e-3 * gw_max
        _close(p1.grad.cpu().numpy(), p2.grad.numpy(), err_msg=n1, scale=gw_max if zero_grad_bias else None)


def test_ssd_mb2_depthwise_extras_chain_vs_torch():
    """samples/ssd_mb2_voc.py: use_depthwise extras 1280@10 -> 512@5 -> 256@3 -> 256@2 -> 128@1, every block on libssdk."""
    from single_shot_detection_amd.bf.modules.conv import DepthwiseConv2dBn
    rng = np.random.default_rng(13)
    extras = detector_builder.get_extras([1280], use_depthwise=True, layers=(('s', 512), ('s', 256), ('s', 256), ('s', 128)))
    _randomize(extras, rng)
    ref = nn.ModuleList([nn.Sequential(*[_ref_of(b) for b in blk]) for blk in extras])
    extras = extras.cuda()
    assert any(isinstance(m, DepthwiseConv2dBn) for m in extras.modules())
    x_np = rng.standard_normal((4, 1280, 10, 10), dtype=np.float32)
    xr = torch.from_numpy(x_np).requires_grad_(True)
    xg = torch.from_numpy(x_np).cuda().requires_grad_(True)
    outs_r, outs_g, cr, cg = [], [], xr, xg
    for blk_g, blk_r, want in zip(extras, ref, (5, 3, 2, 1)):
        cr, cg = blk_r(cr), blk_g(cg)
        assert cg.shape[2] == want
        outs_r.append(cr); outs_g.append(cg)
        _close(cg.detach().cpu().numpy(), cr.detach().numpy())
    # 1x1 maps in train-mode BatchNorm over 4 samples are ill-conditioned: differentiate a weighted sum of the first three levels
    gws = [torch.from_numpy(rng.standard_normal(tuple(o.shape), dtype=np.float32)) for o in outs_r[:3]]
    sum((o * g).sum() for o, g in zip(outs_r[:3], gws)).backward()
    sum((o * g.cuda()).sum() for o, g in zip(outs_g[:3], gws)).backward()
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())


@pytest.mark.parametrize('cin,cout,h,k,stride,pad', [(32, 64, 10, 3, 2, 1), (64, 32, 7, 3, 1, 1), (16, 24, 9, 5, 2, 2), (1280, 512, 10, 3, 2, 1)])
def test_depthwise_conv2d_bn_vs_torch(cin, cout, h, k, stride, pad):
    """DepthwiseConv2dBn (bf/modules/conv.py:39-85, the `use_depthwise` extras of samples/ssd_mb2_voc.py) on libssdk vs stock torch CPU."""
    from single_shot_detection_amd.bf.modules.conv import DepthwiseConv2dBn
    rng = np.random.default_rng(31)
    torch.manual_seed(0)
    blk = DepthwiseConv2dBn(cin, cout, kernel_size=k, stride=stride, padding=pad)
    _randomize(blk, rng)
    ref = _RefDepthwise(blk)
    blk = blk.cuda()
    assert blk._hip_ok()
    x_np = rng.standard_normal((4, cin, h, h), dtype=np.float32)
    xr = torch.from_numpy(x_np).requires_grad_(True)
    xg = torch.from_numpy(x_np).cuda().requires_grad_(True)
    yr, yg = ref(xr), blk(xg)
    assert yg.shape == yr.shape
    _close(yg.detach().cpu().numpy(), yr.detach().numpy())
    gw = torch.from_numpy(rng.standard_normal(tuple(yr.shape), dtype=np.float32))
    (yr * gw).sum().backward()
    (yg * gw.cuda()).sum().backward()
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())
    for (n1, p1), (n2, p2) in zip(sorted(blk.named_parameters()), sorted(ref.mods.named_parameters())):
        assert n1 == n2
        scale = float(p2.grad.abs().max()) + 1e-6
        _close(p1.grad.cpu().numpy(), p2.grad.numpy(), err_msg=n1)
    # eval mode (running statistics)
    blk.eval(); ref.eval()
    with torch.no_grad():
        _close(blk(xg).cpu().numpy(), ref(xr).numpy())


def test_conv2dbn_with_syncbatchnorm_keeps_torch_norm():
    """detection.init(distributed=True) converts BatchNorm2d to SyncBatchNorm: the conv stays on libssdk, the norm on torch's kernels."""
    rng = np.random.default_rng(3)
    m = conv.Conv2dBn(32, 64, kernel_size=3, stride=2, padding=1, bias=False)
    _randomize(m, rng)
    ref = _RefConv2dBn(m)
    gpu = nn.SyncBatchNorm.convert_sync_batchnorm(m).cuda()
    assert isinstance(gpu.bn, nn.SyncBatchNorm)
    x = rng.standard_normal((4, 32, 9, 9), dtype=np.float32)
    for train in (True, False):
        gpu.train(train); ref.train(train)
        yg = gpu(torch.from_numpy(x).cuda())
        yr = ref(torch.from_numpy(x))
        _close(yg.detach().cpu().numpy(), yr.detach().numpy())


def test_sync_batchnorm_with_one_rank_is_the_local_batchnorm_bit_for_bit():
    """distributed.convert_sync_batchnorm marks the hot-path BatchNorm2d layers; with a single rank the split path
    (ssdk_batchnorm_stats -> [all-reduce] -> ssdk_batchnorm_apply, and the same in the backward) must reproduce the local
    ssdk_batchnorm_fwd / _bwd exactly: outputs, running statistics, all gradients.  (Compared on the norm alone: the convolution's
    backward in front of it sums with atomics, so a whole block is not bit-reproducible from run to run.)"""
    import copy
    from single_shot_detection_amd import ops
    from single_shot_detection_amd.distributed import convert_sync_batchnorm
    rng = np.random.default_rng(11)
    blk = conv.Conv2dBn(32, 64, kernel_size=3, stride=2, padding=1, bias=False)
    _randomize(blk, rng)
    marked = convert_sync_batchnorm(copy.deepcopy(blk))
    assert type(marked.bn) is nn.BatchNorm2d and ops.sync_group_of(marked.bn) == (None,) and ops.sync_group_of(blk.bn) is None
    for relu in (True, False):
        a, b = copy.deepcopy(blk.bn).cuda().train(), copy.deepcopy(marked.bn).cuda().train()
        b._ssdk_sync_group = (None,)
        x = torch.from_numpy(rng.standard_normal((4, 64, 9, 9), dtype=np.float32)).cuda().contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = ops.batch_norm(xa, a, relu=relu), ops.batch_norm(xb, b, relu=relu)
        assert torch.equal(ya, yb)
        g = torch.from_numpy(rng.standard_normal(tuple(ya.shape), dtype=np.float32)).cuda().contiguous(memory_format=torch.channels_last)
        (ya * g).sum().backward()
        (yb * g).sum().backward()
        assert torch.equal(xa.grad, xb.grad)
        for (n1, p1), (n2, p2) in zip(sorted(a.named_parameters()), sorted(b.named_parameters())):
            assert n1 == n2 and torch.equal(p1.grad, p2.grad), n1
        for (n1, t1), (n2, t2) in zip(sorted(a.named_buffers()), sorted(b.named_buffers())):
            assert n1 == n2 and torch.equal(t1, t2), n1


def _sync_bn_rank(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    from single_shot_detection_amd.distributed import convert_sync_batchnorm
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    d = np.load(os.path.join(out_dir, 'in.npz'))
    from single_shot_detection_amd.detection.modules.predictors import SharedConvPredictor
    torch.manual_seed(0)
    tower = SharedConvPredictor([16, 16], [3, 3], 5, False, num_layers=2, num_channels=16)
    tower.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(out_dir, 'state.npz')).items()})
    tower = convert_sync_batchnorm(tower).cuda().train()
    half = slice(rank * 2, rank * 2 + 2)
    xs = [torch.from_numpy(d[f'x{i}'][half]).cuda().requires_grad_(True) for i in range(2)]
    s, l = tower(xs)
    gs = [torch.from_numpy(d[f'g{i}'][half]).cuda() for i in range(4)]
    loss = sum((y * g).sum() for y, g in zip(list(s) + list(l), gs))
    loss.backward()
    out = {f'y{i}': y.detach().cpu().numpy() for i, y in enumerate(list(s) + list(l))}
    out.update({f'dx{i}': x.grad.cpu().numpy() for i, x in enumerate(xs)})
    out.update({'p_' + n: p.grad.cpu().numpy() for n, p in tower.named_parameters()})
    out.update({'b_' + n: b.cpu().numpy() for n, b in tower.named_buffers()})
    np.savez(os.path.join(out_dir, f'out{rank}.npz'), **out)
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_one_process_on_the_whole_batch(tmp_path):
    """Two ranks (both on this one GPU, gloo between them), half the batch each, through a RetinaNet tower whose per-level norms are
    synchronised (one packed all-reduce per tower layer) == the same tower in ONE process on the whole batch with torch's own
    BatchNorm2d on the CPU: outputs, running statistics, input gradients; parameter gradients sum over the ranks."""
    import socket
    import torch.multiprocessing as mp
    from single_shot_detection_amd.detection.modules.predictors import SharedConvPredictor
    rng = np.random.default_rng(5)
    torch.manual_seed(0)
    tower = SharedConvPredictor([16, 16], [3, 3], 5, False, num_layers=2, num_channels=16)
    with torch.no_grad():
        for n, p in tower.named_parameters():
            p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape), dtype=np.float32) * (0.2 if p.dim() > 1 else 0.5)) + (1.0 if n.endswith('weight') and p.dim() == 1 else 0.0))
    np.savez(tmp_path / 'state.npz', **{k: v.numpy() for k, v in tower.state_dict().items()})
    data = {'x0': rng.standard_normal((4, 16, 6, 6), dtype=np.float32), 'x1': rng.standard_normal((4, 16, 3, 3), dtype=np.float32)}
    for i, hw in enumerate((6, 3, 6, 3)):
        data[f'g{i}'] = rng.standard_normal((4, 16, hw, hw), dtype=np.float32)
    np.savez(tmp_path / 'in.npz', **data)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sync_bn_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # reference: stock torch modules on the CPU, whole batch (predictors.py:60-76: conv -> ReLU -> per-level BatchNorm2d)
    ref = SharedConvPredictor([16, 16], [3, 3], 5, False, num_layers=2, num_channels=16)
    ref.load_state_dict(tower.state_dict())
    ref.train()
    xs = [torch.from_numpy(data[f'x{i}']).requires_grad_(True) for i in range(2)]
    ss, ls = list(xs), list(xs)
    for sc, lc, sn, ln in zip(ref.convs['score'], ref.convs['loc'], ref.norms['score'], ref.norms['loc']):
        ss = [n(torch.relu(sc.conv(x))) for n, x in zip(sn, ss)]
        ls = [n(torch.relu(lc.conv(x))) for n, x in zip(ln, ls)]
    sum((y * torch.from_numpy(data[f'g{i}'])).sum() for i, y in enumerate(ss + ls)).backward()
    outs = [np.load(tmp_path / f'out{r}.npz') for r in range(2)]
    for i, y in enumerate(ss + ls):
        got = np.concatenate([outs[0][f'y{i}'], outs[1][f'y{i}']], 0)
        _close(got, y.detach().numpy())
    for i, x in enumerate(xs):
        got = np.concatenate([outs[0][f'dx{i}'], outs[1][f'dx{i}']], 0)
        _close(got, x.grad.numpy())
    for n, p in ref.named_parameters():
        got = outs[0]['p_' + n] + outs[1]['p_' + n]
        _close(got, p.grad.numpy(), err_msg=n)
    for n, b in ref.named_buffers():
        for r in range(2):
            np.testing.assert_allclose(outs[r]['b_' + n], b.numpy(), rtol=1e-4, atol=1e-5, err_msg=n)


def test_deferred_weight_gradients_equal_immediate_ones():
    """ops.defer_weight_gradients: the pyramid tail's weight gradients computed in one grouped launch at the end of the backward pass
    (written into param.grad directly) are the gradients of the immediate path; a second backward accumulates like autograd does."""
    import copy
    from single_shot_detection_amd import ops
    rng = np.random.default_rng(21)
    extras = detector_builder.get_extras([512], layers=(('s', 512), ('s', 256), ('s', 256), ('s', 256)))
    _randomize(extras, rng)
    a, b = copy.deepcopy(extras).cuda(), copy.deepcopy(extras).cuda()
    x_np = rng.standard_normal((2, 512, 18, 18), dtype=np.float32)

    def run(mod, defer, times=1):
        prev = ops.defer_weight_gradients(defer)
        try:
            for _ in range(times):
                x = torch.from_numpy(x_np).cuda().requires_grad_(True)
                y, outs = x, []
                for blk in mod:
                    y = blk(y)
                    outs.append(y)
                sum((o * o).sum() for o in outs).backward()
        finally:
            ops.defer_weight_gradients(prev)
        return x.grad

    ga, gb = run(a, False, 2), run(b, True, 2)
    np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(ga.abs().max()))
    for (n1, p1), (n2, p2) in zip(sorted(a.named_parameters()), sorted(b.named_parameters())):
        assert n1 == n2 and p2.grad is not None, n1
        np.testing.assert_allclose(p2.grad.cpu().numpy(), p1.grad.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(p1.grad.abs().max()) + 1e-7, err_msg=n1)
    assert not ops._pending_wgrads


def test_deferred_weight_gradients_survive_a_backward_pass_that_raised():
    """A backward pass that raises drops autograd's end-of-pass callbacks: the deferred jobs it queued must neither block the next pass's
    flush nor be added into the next step's gradients; a weight with a tensor hook is never deferred (the hook must fire)."""
    import copy
    from single_shot_detection_amd import ops
    rng = np.random.default_rng(23)
    extras = detector_builder.get_extras([512], layers=(('s', 512), ('s', 256)))
    _randomize(extras, rng)
    a, b = copy.deepcopy(extras).cuda(), copy.deepcopy(extras).cuda()
    x_np = rng.standard_normal((2, 512, 18, 18), dtype=np.float32)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')

    def step(mod, fail=False):
        x = torch.from_numpy(x_np).cuda().requires_grad_(True)
        y, outs = x, []
        for i, blk in enumerate(mod):
            if fail and i == 0:
                y = Boom.apply(y)   # (raises after the second block's convolutions have queued their jobs)
            y = blk(y)
            outs.append(y)
        sum((o * o).sum() for o in outs).backward()

    step(a)
    fired = []
    prev = ops.defer_weight_gradients(True)
    try:
        with pytest.raises(RuntimeError, match='boom'):
            step(b, fail=True)
        assert ops._pending_wgrads, 'the failed pass was expected to leave deferred jobs behind'
        for p in b.parameters():
            p.grad = None
        hooked = [p for p in b.parameters() if p.dim() == 4][0]
        hooked.register_hook(lambda g: fired.append(1) or g)
        step(b)
    finally:
        ops.defer_weight_gradients(prev)
    assert fired and not ops._pending_wgrads
    for (n1, p1), (n2, p2) in zip(sorted(a.named_parameters()), sorted(b.named_parameters())):
        assert n1 == n2 and p2.grad is not None, n1
        np.testing.assert_allclose(p2.grad.cpu().numpy(), p1.grad.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(p1.grad.abs().max()) + 1e-7, err_msg=n1)


def test_prepared_weight_transposes_equal_the_per_call_ones_and_never_go_stale():
    """ops.prepare_weight_transposes: one grouped re-layout launch for a chain's backward-data GEMMs (ssdk_conv2d_transpose_weights +
    ssdk_conv_desc::w_t).  Same gradients as when every backward call re-lays out its own weights; and a layout prepared BEFORE the
    weights changed (an optimizer step between two forward passes) is not used for the new weights."""
    import copy
    from single_shot_detection_amd import ops
    rng = np.random.default_rng(33)
    extras = detector_builder.get_extras([512], layers=(('s', 512), ('s', 256), ('s', 256)))
    _randomize(extras, rng)
    a, b = copy.deepcopy(extras).cuda(), copy.deepcopy(extras).cuda()
    x_np = rng.standard_normal((2, 512, 18, 18), dtype=np.float32)

    def run(mod, prepare):
        x = torch.from_numpy(x_np).cuda().requires_grad_(True)
        if prepare:
            assert ops.prepare_weight_transposes(mod) == 6        # the six convolutions of the three blocks, one launch
            assert ops.prepare_weight_transposes(mod) == 6        # ... every call: a layout of an earlier step is never trusted
        y, outs = x, []
        for blk in mod:
            y = blk(y)
            outs.append(y)
        sum((o * o).sum() for o in outs).backward()
        return x.grad

    ga, gb = run(a, False), run(b, True)
    np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(ga.abs().max()))
    # the weights change in place (what an optimizer step does): the prepared layouts are now those of OLD weights
    with torch.no_grad():
        for m in (a, b):
            for p in m.parameters():
                if p.dim() == 4:
                    p.mul_(-0.5)
    for m in (a, b):
        m.zero_grad(set_to_none=True)
    conv0 = b[0][0].conv
    assert ops._transposed_weights_of(conv0.weight, conv0.stride[0]) is None   # (and the backward pass that used a layout forgot it)
    ga2, gb2 = run(a, False), run(b, False)   # b is NOT prepared again: its backward must re-lay out the new weights itself
    np.testing.assert_allclose(gb2.cpu().numpy(), ga2.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(ga2.abs().max()))
    assert not np.allclose(gb2.cpu().numpy(), gb.cpu().numpy(), rtol=1e-3, atol=1e-5 * float(ga.abs().max()))


@pytest.mark.parametrize('cin,cout,k,stride,pad,hw,batch', [(64, 128, 3, 1, 1, 64, 8),     # 256 row tiles: not split over K -> statistics in the epilogue
                                                             (32, 256, 1, 1, 0, 48, 16),    # two column blocks, 1 x 1
                                                             (64, 64, 3, 2, 1, 9, 2)])      # small map: split over K -> the pass of its own
def test_batchnorm_statistics_from_the_conv_epilogue_equal_the_separate_pass(cin, cout, k, stride, pad, hw, batch, monkeypatch):
    """ssdk_conv_desc::stats / ops.conv2d_batch_norm: Conv2dBn whose BatchNorm statistics are accumulated by the convolution's epilogue
    (or, for a split-K convolution, by the library's own pass after it) against the same block with conv and norm run apart: outputs,
    running statistics, and every gradient."""
    import copy
    from single_shot_detection_amd import ops
    rng = np.random.default_rng(5)
    blk = conv.Conv2dBn(cin, cout, k, stride=stride, padding=pad, bias=True)
    _randomize(blk, rng)
    a, b = copy.deepcopy(blk).cuda().train(), copy.deepcopy(blk).cuda().train()
    x_np = rng.standard_normal((batch, cin, hw, hw), dtype=np.float32)

    def run(mod):
        x = torch.from_numpy(x_np).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = mod(x)
        (y * y).sum().backward()
        return y.detach(), x.grad

    before = ops.fused_stats_calls
    ya, ga = run(a)
    assert ops.fused_stats_calls == before + 1
    monkeypatch.setattr(ops, '_local_training_chain', lambda bn, device: None)
    yb, gb = run(b)
    assert ops.fused_stats_calls == before + 1
    scale = float(yb.abs().max())
    np.testing.assert_allclose(ya.cpu().numpy(), yb.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
    _close(ga.cpu().numpy(), gb.cpu().numpy())
    for name in ('running_mean', 'running_var'):
        np.testing.assert_allclose(getattr(a.bn, name).cpu().numpy(), getattr(b.bn, name).cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert int(a.bn.num_batches_tracked) == int(b.bn.num_batches_tracked) == 1
    gmax = max(float(p.grad.abs().max()) for p in b.parameters())   # (the conv bias' gradient through a BatchNorm is zero up to rounding: one scale for all)
    for (n1, p1), (n2, p2) in zip(sorted(a.named_parameters()), sorted(b.named_parameters())):
        _close(p1.grad.cpu().numpy(), p2.grad.cpu().numpy(), err_msg=n1, scale=gmax if n1 == 'conv.bias' else None)


@pytest.mark.parametrize('cin,cout,k,stride,pad,sizes,B', [
    (256, 512, 3, 2, 1, (19,), 4),            # the SSD tail's strided 3 x 3: the gather form of its data gradient
    (512, 128, 1, 1, 0, (9,), 4),             # 1 x 1
    (128, 256, 3, 2, 1, (5,), 8),             # small map (a K split with atomics in the default mode)
    (256, 256, 3, 1, 1, (16, 8, 4), 2),       # one weight tensor over three maps (shared dw / db: reduced descriptor by descriptor)
    (64, 40, 3, 2, 0, (9,), 3),               # no padding, Cout not a multiple of 32
])
def test_deterministic_mode_convolution_vs_torch_cpu(cin, cout, k, stride, pad, sizes, B):
    """ops.conv2d under ops.deterministic(): forward, data / weight / bias gradients against torch's fp32 CPU convolution, and the same
    call twice gives the same bits (no fp32 atomics: the strided data gradient in its output-stationary form, K-split weight-gradient
    copies and per-workgroup bias column sums added in a fixed order)."""
    import torch.nn.functional as F
    from single_shot_detection_amd import ops
    torch.manual_seed(13)
    w = (torch.randn((cout, cin, k, k), device='cuda') * 0.03).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = (torch.randn((cout,), device='cuda') * 0.1).requires_grad_(True)
    xs = [torch.randn((B, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for h in sizes]
    runs = []
    with ops.deterministic():
        for _ in range(2):
            ys = ops.conv2d(xs, w, b, stride, pad)
            gs = [torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5 + i)) for i, y in enumerate(ys)]
            grads = torch.autograd.grad(ys, [w, b] + xs, gs)
            runs.append(([y.detach() for y in ys], grads))
    for t0, t1 in zip(runs[0][0] + list(runs[0][1]), runs[1][0] + list(runs[1][1])):
        assert torch.equal(t0, t1)
    wc, bc = w.detach().cpu().contiguous().requires_grad_(True), b.detach().cpu().requires_grad_(True)
    xc = [x.detach().cpu().contiguous().requires_grad_(True) for x in xs]
    yc = [F.conv2d(x, wc, bc, stride=stride, padding=pad) for x in xc]
    gc = torch.autograd.grad(yc, [wc, bc] + xc, [g.cpu() for g in gs])
    for y, r in zip(runs[1][0], yc):
        assert float((y.cpu() - r.detach()).abs().max()) <= 2e-6 * (k * k * cin) ** 0.5 + 1e-5 * float(r.detach().abs().max())
    for a, r in zip(runs[1][1], gc):
        scale = float(r.abs().max()) + 1e-12
        assert float((a.cpu() - r).abs().max()) <= 2e-5 * scale + 1e-6, (tuple(r.shape), float((a.cpu() - r).abs().max()), scale)


def test_prepared_weight_transposes_follow_a_fused_optimizer_step():
    """ops.prepare_weight_transposes caches the weights' backward-data layouts per parameter.  torch.optim.SGD(fused=True) updates a
    parameter WITHOUT bumping its version counter, so a cache keyed on the version served the layout of an OLD step's weights to every
    later backward pass (round 3; found by the deterministic-mode test).  Two steps with a large learning rate: the second step's input
    gradient must be the one torch's CPU convolution gives for the UPDATED weights."""
    import torch.nn.functional as F
    from single_shot_detection_amd import ops
    from single_shot_detection_amd.bf.modules.conv import Conv2dBn
    torch.manual_seed(3)
    tail = torch.nn.Sequential(Conv2dBn(64, 32, kernel_size=1, bias=False, use_bn=False, activation_params=None),
                               Conv2dBn(32, 64, kernel_size=3, stride=2, padding=1, bias=False, use_bn=False, activation_params=None)).cuda()
    tail = tail.to(memory_format=torch.channels_last).train()
    try:
        opt = torch.optim.SGD(tail.parameters(), lr=0.5, fused=True)
    except (TypeError, RuntimeError, ValueError):
        pytest.skip('no fused SGD in this torch build')
    x = torch.randn((2, 64, 9, 9), device='cuda').contiguous(memory_format=torch.channels_last)
    for step in range(3):
        xi = x.clone().requires_grad_(True)
        opt.zero_grad(set_to_none=True)
        assert ops.prepare_weight_transposes(tail) == 2      # (every call re-lays out every weight)
        y = tail(xi)
        g = torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(step))
        y.backward(g)
        w0, w1 = (m.conv.weight.detach().cpu().contiguous() for m in tail)
        xc = x.cpu().contiguous().requires_grad_(True)
        F.conv2d(F.conv2d(xc, w0), w1, stride=2, padding=1).backward(g.cpu())
        scale = float(xc.grad.abs().max())
        assert float((xi.grad.cpu() - xc.grad).abs().max()) <= 2e-5 * scale, (step, float((xi.grad.cpu() - xc.grad).abs().max()), scale)
        opt.step()   # (moves the weights by half their gradient: a stale layout is off by far more than the bound)


def test_prepared_weight_transposes_do_not_survive_an_optimizer_step():
    """Advisor (round 4): a prepared layout that no backward pass consumed (forward with grad enabled, then no backward) must not serve a
    LATER step's backward pass once a fused optimizer step has changed the weights without bumping ``_version`` -- every optimizer step
    drops all prepared layouts (ops._forget_all_transposed_weights); the next backward re-lays out its own weights."""
    import torch.nn.functional as F
    from single_shot_detection_amd import ops
    from single_shot_detection_amd.bf.modules.conv import Conv2dBn
    torch.manual_seed(5)
    tail = torch.nn.Sequential(Conv2dBn(64, 32, kernel_size=1, bias=False, use_bn=False, activation_params=None)).cuda()
    tail = tail.to(memory_format=torch.channels_last).train()
    try:
        opt = torch.optim.SGD(tail.parameters(), lr=0.5, fused=True)
    except (TypeError, RuntimeError, ValueError):
        pytest.skip('no fused SGD in this torch build')
    x = torch.randn((2, 64, 9, 9), device='cuda').contiguous(memory_format=torch.channels_last)
    xi = x.clone().requires_grad_(True)
    tail(xi).sum().backward()                  # gives the weights a gradient
    assert ops.prepare_weight_transposes(tail) == 1
    tail(x.clone().requires_grad_(True))       # a forward pass with grad enabled and NO backward pass: the prepared layout stays behind
    assert len(ops._wt_cache) == 1
    opt.step()                                 # fused: the weights change, their version counter does not
    assert len(ops._wt_cache) == 0
    xi = x.clone().requires_grad_(True)
    y = tail(xi)                               # (no prepare_weight_transposes in front of this forward pass)
    g = torch.randn(y.shape, device='cuda')
    y.backward(g)
    w0 = tail[0].conv.weight.detach().cpu().contiguous()
    xc = x.cpu().contiguous().requires_grad_(True)
    F.conv2d(xc, w0).backward(g.cpu())
    scale = float(xc.grad.abs().max())
    assert float((xi.grad.cpu() - xc.grad).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize('B,C,H', [(2, 1024, 64), (3, 40, 5), (1, 128, 17), (4, 256, 2), (2, 64, 1)])
def test_global_avg_pool_and_sigmoid_gate_vs_torch(B, C, H):
    """ops.global_avg_pool / ops.sigmoid_gate (M2Det SFAM, bf/modules/features.py:290-298) against torch on shapes that exercise the
    kernels' row loops (four rows in flight, then single rows; HW = 1) and column tails (C / 4 not a multiple of the 16-column block)."""
    import torch.nn.functional as F
    from single_shot_detection_amd import ops
    torch.manual_seed(B * 1000 + C + H)
    x = torch.randn((B, C, H, H), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    z = torch.randn((B, C, 1, 1), device='cuda').requires_grad_(True)
    g = torch.randn((B, C, H, H), device='cuda').contiguous(memory_format=torch.channels_last)
    pooled = ops.global_avg_pool(x)
    ref = F.adaptive_avg_pool2d(x.detach().double(), 1)
    assert float((pooled.detach().double() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max())) + 1e-6
    (gx,) = torch.autograd.grad(pooled, [x], torch.ones_like(pooled))
    assert float((gx - 1.0 / (H * H)).abs().max()) <= 1e-7
    out = ops.sigmoid_gate(x, z)
    xr, zr = x.detach().double().requires_grad_(True), z.detach().double().requires_grad_(True)
    outr = xr * torch.sigmoid(zr)
    assert float((out.detach().double() - outr.detach()).abs().max()) <= 1e-5
    dx, dz = torch.autograd.grad(out, [x, z], g)
    dxr, dzr = torch.autograd.grad(outr, [xr, zr], g.double())
    assert float((dx.double() - dxr).abs().max()) <= 1e-5
    assert float((dz.double() - dzr).abs().max()) <= 2e-5 * float(dzr.abs().max()) + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('B,n_pieces,Cp,sizes', [(2, 8, 16, (9, 5, 3, 1)), (3, 3, 8, (17, 4)), (1, 1, 32, (6,)), (4, 8, 128, (8, 2))])
def test_sfam_over_pieces_vs_torch(B, n_pieces, Cp, sizes):
    """ops.sfam_pieces (M2Det SFAM without the concatenated maps: ssdk_sfam_pool_fwd / gate_fwd / gate_bwd_reduce / gate_bwd_apply + the
    grouped 1 x 1 fc convolutions) against the reference's arithmetic in torch fp64 -- torch.cat, adaptive_avg_pool2d, fc1 + relu, fc2,
    x * sigmoid (bf/modules/features.py:385, :286-298) -- outputs, every piece's gradient, the fc weights' and biases' gradients; and the
    module's own torch.cat path (SSDK_SFAM_CAT=1) gives the same numbers."""
    import os
    import torch.nn.functional as F
    from single_shot_detection_amd import ops
    from single_shot_detection_amd.bf.modules.features import ScalewiseFeatureAggregationModule
    torch.manual_seed(B * 100 + n_pieces * 10 + Cp)
    C = n_pieces * Cp
    sfam = ScalewiseFeatureAggregationModule(C, len(sizes), reduction_ratio=2).cuda()
    pieces = [[torch.randn((B, Cp, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(n_pieces)]
              for h in sizes]
    gouts = [torch.randn((B, C, h, h), device='cuda').contiguous(memory_format=torch.channels_last) for h in sizes]
    assert ops.sfam_pieces_ok(pieces, list(sfam.fc1), list(sfam.fc2))
    params = [p for p in sfam.parameters()]
    flat = [p for ps in pieces for p in ps]

    def run():
        outs = sfam.forward_pieces(pieces)
        grads = torch.autograd.grad(outs, flat + params, gouts)
        return [o.detach() for o in outs], grads
    outs, grads = run()
    # reference arithmetic, fp64
    ref_pieces = [[p.detach().double().requires_grad_(True) for p in ps] for ps in pieces]
    ref_params = [p.detach().double().requires_grad_(True) for p in params]
    n = len(sizes)
    ref_outs = []
    for s in range(n):
        f = torch.cat(ref_pieces[s], dim=1)
        # (ModuleList order of the parameters: fc1.0.weight, fc1.0.bias, fc1.1.weight, ..., then fc2.*)
        w1, b1, w2, b2 = ref_params[2 * s], ref_params[2 * s + 1], ref_params[2 * n + 2 * s], ref_params[2 * n + 2 * s + 1]
        x = F.adaptive_avg_pool2d(f, 1)
        x = F.relu(F.conv2d(x, w1, b1))
        x = F.conv2d(x, w2, b2)
        ref_outs.append(f * torch.sigmoid(x))
    ref_grads = torch.autograd.grad(ref_outs, [p for ps in ref_pieces for p in ps] + ref_params, [g.double() for g in gouts])
    for o, r in zip(outs, ref_outs):
        assert float((o.double() - r.detach()).abs().max()) <= 2e-5 * max(1.0, float(r.detach().abs().max()))
    for i, (g, r) in enumerate(zip(grads, ref_grads)):
        assert g.shape == r.shape
        assert float((g.double() - r).abs().max()) <= 5e-5 * max(1.0, float(r.abs().max())), i
    # the torch.cat path of the module (the pre-round-4 form) agrees
    os.environ['SSDK_SFAM_CAT'] = '1'
    try:
        outs2, grads2 = run()
    finally:
        del os.environ['SSDK_SFAM_CAT']
    for a, b in zip(outs + list(grads), outs2 + list(grads2)):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_streamk_data_gradient_equals_the_whole_tile_launch():
    """ssdk_conv2d_bwd_sk with SSDK_CONV_STREAMK_BWD=1 (opt-in, read once per process: a child process): a stride-1 data gradient of a few
    rounds of tiles whose last round is partly filled (576 row tiles x 2 column blocks = 1 152 tiles on 512 slots, the shape class of the
    RetinaNet towers) runs as igemm_streamk_kernel<true>; same numbers as the whole-tile launch (SSDK_CONV_NO_STREAMK=1) up to the
    summation order inside a tile cut in two, and as torch's own data gradient."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSDK_CONV_STREAMK_BWD='1')
    out = subprocess.run([sys.executable, os.path.join(repo, 'tests', 'streamk_bwd_worker.py')], cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res['timeouts'] == 0
    assert res['differs']                       # (the two launches do differ: some tiles are summed in two parts)
    assert res['max_vs_plain'] <= 1e-5          # (relative to the largest gradient: sums of 2 304 products, cut at another place)
    assert res['dw_max_vs_plain'] <= 1e-4
    assert res['max_vs_torch'] <= 2e-4          # (MIOpen's own algorithm: a looser bound)
