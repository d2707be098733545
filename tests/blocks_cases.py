"""The conv-BN-ReLU compositions of the hot path as seeded cases that run through ANY implementation of the
reference's module interface: ``tools/gen_golden.py --only blocks`` runs them through the reference's own modules on the CPU (build
container only) and writes ``tests/golden/blocks_small.npz``; ``tests/test_blocks_golden_gpu.py`` runs the same cases through this
repository's modules on the GPU and compares.  What a case pins: op order (conv -> BN -> ReLU, bf/modules/conv.py:30-36,72-85; conv ->
ReLU -> per-level BN, detection/modules/predictors.py:60-76; the FPN / TUM / SFAM graphs, bf/modules/features.py:103-117,215-300), the
state_dict names (the parameters are filled by name), BatchNorm momentum / unbiased running variance / num_batches_tracked after one
train() step, and eval() mode.  This file holds no reference code: constructor arguments, shapes and seeds only."""
import zlib

import numpy as np
import torch
import torch.nn as nn

SAMPLE_ABOVE = 10000   # arrays larger than this are stored as N_SAMPLES sampled entries + their fp64 sum and L2 norm
N_SAMPLES = 2048


def fill_module_(module, seed):
    """Deterministic parameters and BatchNorm buffers, by state_dict name (sorted), from numpy PCG64: the same bytes for the reference's
    module and for ours -- provided the names agree, which is part of the contract."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            shape = tuple(p.shape)
            if p.dim() > 1:
                fan_in = int(np.prod(shape[1:]))
                v = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
            elif name.endswith('weight'):   # BatchNorm gamma
                v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            else:                           # biases, BatchNorm beta
                v = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1))
            p.copy_(torch.from_numpy(v).to(p.device))
        for name, b in sorted(module.named_buffers()):
            if name.endswith('running_mean'):
                b.copy_(torch.from_numpy((rng.standard_normal(tuple(b.shape), dtype=np.float32) * np.float32(0.1))).to(b.device))
            elif name.endswith('running_var'):
                b.copy_(torch.from_numpy(rng.uniform(0.5, 2.0, tuple(b.shape)).astype(np.float32)).to(b.device))
            elif name.endswith('num_batches_tracked'):
                b.fill_(3)
    return module


class _StubBase(nn.Module):
    """A three-tap stand-in backbone (stock torch on both sides): only its taps' channel counts and strides matter to the necks."""

    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, 16, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(16, 24, 3, stride=2, padding=1), nn.ReLU(),
                                      nn.Conv2d(24, 40, 3, stride=2, padding=1))


def _chain(modules, xs):
    x, outs = xs[0], []
    for m in modules:
        x = m(x)
        outs.append(x)
    return outs


# name -> (build(mods) -> module, input shapes, forward(module, xs) -> tensor / nested lists of tensors)
# `mods` offers: Conv2dBn, DepthwiseConv2dBn, get_extras, SharedConvPredictor, FeaturePyramid, ThinnedUshapeModule,
# ScalewiseFeatureAggregationModule (the reference's classes in the generator, this repository's in the test)
CASES = {
    'conv2dbn_1x1': (lambda m: m.Conv2dBn(64, 32, kernel_size=1), [(4, 64, 9, 9)], lambda mod, xs: mod(xs[0])),
    'conv2dbn_3x3s2': (lambda m: m.Conv2dBn(32, 64, kernel_size=3, stride=2, padding=1), [(4, 32, 9, 9)], lambda mod, xs: mod(xs[0])),
    'conv2dbn_3x3_nopad': (lambda m: m.Conv2dBn(32, 48, kernel_size=3), [(3, 32, 5, 5)], lambda mod, xs: mod(xs[0])),
    'depthwise_3x3s2': (lambda m: m.DepthwiseConv2dBn(32, 64, kernel_size=3, stride=2, padding=1), [(3, 32, 10, 10)], lambda mod, xs: mod(xs[0])),
    # detector_builder.get_extras of samples/ssd_300_vgg16_voc.py at its own shapes (512 @ 18 x 18 -> 9 -> 5 -> 3 -> 2)
    'extras_ssd300': (lambda m: m.get_extras([512], layers=(('s', 512), ('s', 256), ('s', 256), ('s', 256))), [(2, 512, 18, 18)], _chain),
    # the same builder with use_depthwise (samples/ssd_mb2_voc.py), small
    'extras_depthwise': (lambda m: m.get_extras([64], use_depthwise=True, layers=(('s', 64), ('s', 32))), [(3, 64, 10, 10)], _chain),
    # RetinaNet tower: 2 layers, 5 levels, shared convolutions, per-level norms
    'tower': (lambda m: m.SharedConvPredictor([64] * 5, [9] * 5, 8, False, num_layers=2, num_channels=64),
              [(4, 64, 13, 13), (4, 64, 7, 7), (4, 64, 4, 4), (4, 64, 2, 2), (4, 64, 1, 1)], lambda mod, xs: mod(xs)),
    'fpn': (lambda m: m.FeaturePyramid(_StubBase(), (1, 3, 4), pyramid_layers=5, pyramid_channels=32), [(2, 3, 64, 64)],
            lambda mod, xs: mod(xs[0])[0]),
    'tum': (lambda m: m.ThinnedUshapeModule(in_channels=48, inner_channels=32, out_channels=16, num_scales=4), [(2, 48, 16, 16)],
            lambda mod, xs: mod(xs[0])),
    'sfam': (lambda m: m.ScalewiseFeatureAggregationModule(num_channels=32, num_scales=3, reduction_ratio=4),
             [(2, 32, 8, 8), (2, 32, 4, 4), (2, 32, 2, 2)], lambda mod, xs: mod(xs)),
}


def case_seed(name):
    return zlib.crc32(name.encode()) % 100000


def case_inputs(name):
    rng = np.random.default_rng(case_seed(name) + 1)
    return [rng.standard_normal(shape, dtype=np.float32) for shape in CASES[name][1]]


def _flatten(o):
    if isinstance(o, torch.Tensor):
        return [o]
    out = []
    for e in o:
        out += _flatten(e)
    return out


def pack(key, arr, res):
    """Store an array whole, or -- above SAMPLE_ABOVE elements -- as N_SAMPLES entries at seeded positions plus two checksums."""
    a = np.ascontiguousarray(arr)
    if a.size <= SAMPLE_ABOVE:
        res[key] = a
        return
    idx = np.random.default_rng(zlib.crc32(key.encode())).choice(a.size, N_SAMPLES, replace=False)
    res[key + '__samples'] = a.reshape(-1)[idx]
    res[key + '__sum_l2'] = np.array([a.astype(np.float64).sum(), np.sqrt((a.astype(np.float64) ** 2).sum())])
    res[key + '__shape'] = np.array(a.shape, np.int64)


def run_case(name, mods, device):
    """Build, fill, run: eval() forward + backward, then ONE train() forward + backward; returns {key: array} (see pack)."""
    build, _, forward = CASES[name]
    torch.manual_seed(0)
    module = fill_module_(build(mods), case_seed(name)).to(device)
    xs_np = case_inputs(name)
    res = {}
    for mode in ('eval', 'train'):
        module.train(mode == 'train')
        module.zero_grad(set_to_none=True)
        xs = [torch.from_numpy(x).to(device).requires_grad_(True) for x in xs_np]
        ys = _flatten(forward(module, xs))
        grng = np.random.default_rng(case_seed(name) + 7)
        gs = [torch.from_numpy(grng.standard_normal(tuple(y.shape), dtype=np.float32)).to(device) for y in ys]
        torch.autograd.backward(ys, gs)
        for i, y in enumerate(ys):
            pack(f'{name}/{mode}/y{i}', y.detach().cpu().numpy(), res)
        for i, x in enumerate(xs):
            pack(f'{name}/{mode}/dx{i}', x.grad.detach().cpu().numpy(), res)
        for pname, p in sorted(module.named_parameters()):
            if pname.startswith('base.'):
                continue   # (the stub backbone is stock torch on both sides)
            assert p.grad is not None, (name, mode, pname)
            pack(f'{name}/{mode}/dp/{pname}', p.grad.detach().cpu().numpy(), res)
    for bname, b in sorted(module.named_buffers()):   # after the one train() step: momentum, unbiased variance, the step counter
        res[f'{name}/buffers/{bname}'] = b.detach().cpu().numpy()
    return res
