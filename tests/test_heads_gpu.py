"""GPU parity: multi-scale head convolutions (H1) through the C ABI -- layout pinned by the reference's own
Predictor.forward golden (tests/golden/heads_small.npz), arithmetic against torch's fp32 CPU convolution.
fp32 MFMA accumulates in a different order than the CPU kernel: tolerances are relative to the K-reduction size."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from single_shot_detection_amd.detection import detector_builder
from single_shot_detection_amd.detection.modules.heads import multi_level_heads
from conftest import GOLDEN
import os

pytestmark = pytest.mark.gpu


def build_heads(levels, num_classes, weights, dev='cuda'):
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], num_classes)
    for i, head in enumerate(heads):
        for kind in ('score', 'loc'):
            w, b = weights[(kind, i)]
            with torch.no_grad():
                head[kind].weight.copy_(torch.from_numpy(w))
                head[kind].bias.copy_(torch.from_numpy(b))
    return heads.to(dev)


def reference_forward(xs, weights, levels):
    """detector.py:50-66 with torch CPU convs."""
    scores, locs = [], []
    for i, x in enumerate(xs):
        ws, bs = weights[('score', i)]
        wl, bl = weights[('loc', i)]
        s = F.conv2d(x, torch.from_numpy(ws), torch.from_numpy(bs), padding=1)
        l = F.conv2d(x, torch.from_numpy(wl), torch.from_numpy(bl), padding=1)
        scores.append(s.permute(0, 2, 3, 1).contiguous().view(x.size(0), -1))
        locs.append(l.permute(0, 2, 3, 1).contiguous().view(x.size(0), -1))
    return torch.cat(scores, 1), torch.cat(locs, 1)


def test_heads_layout_vs_reference_golden():
    g = np.load(os.path.join(GOLDEN, 'heads_small.npz'))
    levels = [tuple(int(v) for v in r) for r in g['levels']]
    C = int(g['num_classes'])
    weights = {(k, i): (g[f'w_{k}_{i}'], g[f'b_{k}_{i}']) for i in range(len(levels)) for k in ('score', 'loc')}
    heads = build_heads(levels, C, weights)
    xs = [torch.from_numpy(g[f'x_{i}']).cuda().requires_grad_(True) for i in range(len(levels))]
    scores, locs = multi_level_heads(xs, xs, heads)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), g['scores'], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(locs.detach().cpu().numpy(), g['locs'], rtol=1e-5, atol=2e-5)
    (scores * torch.from_numpy(g['g_scores']).cuda()).sum().add((locs * torch.from_numpy(g['g_locs']).cuda()).sum()).backward()
    for i in range(len(levels)):
        np.testing.assert_allclose(xs[i].grad.cpu().numpy(), g[f'dx_{i}'], rtol=1e-4, atol=2e-5)
        for k in ('score', 'loc'):
            np.testing.assert_allclose(heads[i][k].weight.grad.cpu().numpy(), g[f'dw_{k}_{i}'], rtol=1e-4, atol=5e-5)
            np.testing.assert_allclose(heads[i][k].bias.grad.cpu().numpy(), g[f'db_{k}_{i}'], rtol=1e-4, atol=5e-5)


SSD300_LEVELS = [(512, 37, 4), (512, 18, 6), (512, 9, 6), (256, 5, 6), (256, 3, 4), (256, 2, 4)]
M2DET_LEVELS = [(1024, 64, 4), (1024, 32, 6), (1024, 16, 6), (1024, 8, 6), (1024, 4, 4), (1024, 2, 4)]   # samples/m2det_512_vgg16_coco.py


@pytest.mark.parametrize('levels,C,B', [
    (SSD300_LEVELS, 81, 2),                                                                      # ssd_300_vgg16_voc
    ([(96, 19, 4), (1280, 10, 6), (512, 5, 6), (256, 3, 6), (256, 2, 4), (128, 1, 4)], 21, 2),   # ssd_mb2_voc
    ([(24, 7, 3), (40, 5, 5)], 7, 3),                                                            # Cin % 32 != 0, odd N
    ([(256, 8, 9)], 80, 1),                                                                      # retina level (nb=9, C=80)
    (SSD300_LEVELS, 21, 2),                                                                      # ssd_300 with the VOC class count (N = 100 / 150)
])
@pytest.mark.parametrize('pixel_density,mode', [(1.0, None), (0.05, None), (0.05, '1'), (0.05, '2'), (0.4, '0'), (0.4, '1'), (0.4, '2')])
def test_heads_vs_torch_cpu_conv(levels, C, B, pixel_density, mode, monkeypatch):
    """pixel_density 1.0 exercises the dense backward kernels, < 1 the sparse ones: the device picks the path (0 dense, 1 rows =
    pixels with a gradient: compacted wgrad + scatter dgrad, 2 rows = single anchors with a gradient) from the densities;
    SSDK_HEADS_BWD_MODE forces one so that every form is checked on the same gradients."""
    _heads_vs_torch_cpu_conv(levels, C, B, pixel_density, mode, monkeypatch)


@pytest.mark.parametrize('B,pixel_density,mode', [(2, 1.0, None), (2, 0.05, None), (1, 0.05, '1'), (1, 0.05, '2'), (1, 0.4, '0')])
def test_m2det_heads_vs_torch_cpu_conv(B, pixel_density, mode, monkeypatch):
    """H1 at m2det_512_vgg16_coco's level shapes (Cin = 1024, 64^2 .. 2^2, C = 81; detection/detector.py:50-66 applied to the six
    MLFPN outputs) against torch's fp32 CPU convolution: forward, dense and sampled-gradient backward."""
    _heads_vs_torch_cpu_conv(M2DET_LEVELS, 81, B, pixel_density, mode, monkeypatch)


def _heads_vs_torch_cpu_conv(levels, C, B, pixel_density, mode, monkeypatch):
    if mode is None:
        monkeypatch.delenv('SSDK_HEADS_BWD_MODE', raising=False)
    else:
        monkeypatch.setenv('SSDK_HEADS_BWD_MODE', mode)
    rng = np.random.default_rng(17)
    weights, xs_np = {}, []
    for i, (cin, h, nb) in enumerate(levels):
        xs_np.append(rng.standard_normal((B, cin, h, h), dtype=np.float32))
        for k, nout in (('score', nb * C), ('loc', nb * 4)):
            weights[(k, i)] = (rng.standard_normal((nout, cin, 3, 3), dtype=np.float32) * np.float32(0.02),
                               rng.standard_normal((nout,), dtype=np.float32) * np.float32(0.1))
    xs_cpu = [torch.from_numpy(x).requires_grad_(True) for x in xs_np]
    ws_cpu = {k: (torch.from_numpy(w).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)) for k, (w, b) in weights.items()}
    scores_ref, locs_ref = [], []
    for i, x in enumerate(xs_cpu):
        s = F.conv2d(x, *ws_cpu[('score', i)], padding=1)
        l = F.conv2d(x, *ws_cpu[('loc', i)], padding=1)
        scores_ref.append(s.permute(0, 2, 3, 1).contiguous().view(B, -1))
        locs_ref.append(l.permute(0, 2, 3, 1).contiguous().view(B, -1))
    scores_ref, locs_ref = torch.cat(scores_ref, 1), torch.cat(locs_ref, 1)
    gs = torch.from_numpy(rng.standard_normal(tuple(scores_ref.shape), dtype=np.float32))
    gl = torch.from_numpy(rng.standard_normal(tuple(locs_ref.shape), dtype=np.float32))
    # sparse upstream gradient like the real loss: most entries are exactly zero
    gs[:, (rng.random(gs.shape[1]) < 0.9)] = 0
    if pixel_density < 1.0:   # and whole pixels (all their anchors, scores and locs) carry no gradient at all
        s_off = l_off = 0
        for cin, h, nb in levels:
            keep = torch.from_numpy((rng.random((B, h * h, 1)) < pixel_density).astype(np.float32))
            keep[:, 0] = 1.0  # at least one row per image so that the list is never empty
            # ... and inside a kept pixel only some anchors (like hard-negative mining: rarely more than one per pixel)
            akeep = torch.from_numpy((rng.random((B, h * h, nb, 1)) < 0.4).astype(np.float32)) * keep.view(B, h * h, 1, 1)
            ns, nl = h * h * nb * C, h * h * nb * 4
            gs[:, s_off:s_off + ns] = (gs[:, s_off:s_off + ns].view(B, h * h, nb, C) * akeep).view(B, -1)
            gl[:, l_off:l_off + nl] = (gl[:, l_off:l_off + nl].view(B, h * h, nb, 4) * akeep).view(B, -1)
            s_off += ns
            l_off += nl
    ((scores_ref * gs).sum() + (locs_ref * gl).sum()).backward()

    heads = build_heads(levels, C, weights)
    xs = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs_np]
    scores, locs = multi_level_heads(xs, xs, heads)
    kmax = 9 * max(l[0] for l in levels)
    tol = 2e-6 * np.sqrt(kmax)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), scores_ref.detach().numpy(), rtol=1e-5, atol=tol)
    np.testing.assert_allclose(locs.detach().cpu().numpy(), locs_ref.detach().numpy(), rtol=1e-5, atol=tol)
    ((scores * gs.cuda()).sum() + (locs * gl.cuda()).sum()).backward()
    for i, (cin, h, nb) in enumerate(levels):
        np.testing.assert_allclose(xs[i].grad.cpu().numpy(), xs_cpu[i].grad.numpy(), rtol=1e-4, atol=1e-4)
        for k in ('score', 'loc'):
            wref, bref = ws_cpu[(k, i)]
            scale = float(wref.grad.abs().max()) + 1e-6
            np.testing.assert_allclose(heads[i][k].weight.grad.cpu().numpy(), wref.grad.numpy(), rtol=1e-4, atol=2e-5 * scale + 1e-5)
            np.testing.assert_allclose(heads[i][k].bias.grad.cpu().numpy(), bref.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('levels,C,B', [
    (SSD300_LEVELS, 21, 32),                      # N = 104 (3 tiles + 8 columns) and 152: the grouped stream-K launch
    (SSD300_LEVELS, 21, 8),                       # ... cut several times per tile
    (SSD300_LEVELS, 21, 1),                       # ... split over K, added into zeroed outputs
    ([(64, 19, 1), (128, 10, 6)], 2, 16),         # N = 12 (the 16-column tile alone) and 40 (1 tile + 8 columns)
    ([(256, 8, 9)], 80, 32),                      # retina score + loc: 720 + 36 = 756 = 23 tiles + 20 columns (no remainder tile: unchanged)
])
def test_heads_forward_16_column_remainder_tile(levels, C, B, monkeypatch):
    """A column space that ends in a tile of at most 16 columns (N = 104 of the 21-class heads) computes that tile with
    v_mfma_f32_16x16x1_4b_f32 (conv.hip dma_tile, ConvProblem::half_last) in every forward form: the result must be the one of the
    32-column tiling (SSDK_CONV_NO_HALF_TILE=1; the default form is also held to torch's CPU convolution by the C = 21 cases of
    test_heads_vs_torch_cpu_conv) up to the order of the fp32 sums over K."""
    rng = np.random.default_rng(3)
    weights, xs_np = {}, []
    for i, (cin, h, nb) in enumerate(levels):
        xs_np.append(rng.standard_normal((B, cin, h, h), dtype=np.float32))
        for k, nout in (('score', nb * C), ('loc', nb * 4)):
            weights[(k, i)] = (rng.standard_normal((nout, cin, 3, 3), dtype=np.float32) * np.float32(0.02),
                               rng.standard_normal((nout,), dtype=np.float32) * np.float32(0.1))
    heads = build_heads(levels, C, weights)
    xs = [torch.from_numpy(x).cuda() for x in xs_np]
    with torch.no_grad():
        monkeypatch.delenv('SSDK_CONV_NO_HALF_TILE', raising=False)
        s1, l1 = multi_level_heads(xs, xs, heads)
        s1b, l1b = multi_level_heads(xs, xs, heads)
        monkeypatch.setenv('SSDK_CONV_NO_HALF_TILE', '1')
        s0, l0 = multi_level_heads(xs, xs, heads)
    tol = 2e-6 * np.sqrt(9 * max(l[0] for l in levels))
    np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=1e-5, atol=tol)
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=1e-5, atol=tol)
    if B >= 8 and len(levels) == len(SSD300_LEVELS):   # (stream-K launches do not add atomically: the same call twice is the same bits)
        assert torch.equal(s1, s1b) and torch.equal(l1, l1b)


@pytest.mark.parametrize('flag', ['SSDK_CONV_BK16', 'SSDK_CONV_TN6'])
def test_heads_forward_experimental_variants(monkeypatch, flag):
    """Opt-in instantiations of the LDS-DMA kernel that were measured and not adopted -- 16-float K slices (3 workgroups per CU: +5 %
    cycles at batch 32, n_score is padded to 16) and 192-column workgroups (+6 % cycles) -- must give the same forward as the default."""
    rng = np.random.default_rng(5)
    levels, C, B = [(64, 9, 4), (128, 5, 6), (96, 12, 9)], 21, 3
    weights = {}
    for i, (cin, h, nb) in enumerate(levels):
        for k, nout in (('score', nb * C), ('loc', nb * 4)):
            weights[(k, i)] = (rng.standard_normal((nout, cin, 3, 3), dtype=np.float32) * np.float32(0.05), rng.standard_normal((nout,), dtype=np.float32))
    heads = build_heads(levels, C, weights)
    xs = [torch.from_numpy(rng.standard_normal((B, cin, h, h), dtype=np.float32)).cuda() for cin, h, _ in levels]
    monkeypatch.delenv(flag, raising=False)
    with torch.no_grad():
        s0, l0 = multi_level_heads(xs, xs, heads)
        monkeypatch.setenv(flag, '1')
        s1, l1 = multi_level_heads(xs, xs, heads)
    np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_heads_accept_nchw_and_channels_last_sources():
    rng = np.random.default_rng(3)
    levels, C, B = [(64, 6, 4)], 5, 2
    weights = {('score', 0): (rng.standard_normal((20, 64, 3, 3), dtype=np.float32) * 0.05, np.zeros(20, np.float32)),
               ('loc', 0): (rng.standard_normal((16, 64, 3, 3), dtype=np.float32) * 0.05, np.zeros(16, np.float32))}
    heads = build_heads(levels, C, weights)
    x = torch.from_numpy(rng.standard_normal((B, 64, 6, 6), dtype=np.float32)).cuda()
    a = multi_level_heads([x], [x], heads)
    b = multi_level_heads([x.contiguous(memory_format=torch.channels_last)] * 1, [x.contiguous(memory_format=torch.channels_last)] * 1, heads)
    # (not bit for bit: at this size the K slices are split over workgroups that add atomically, in any order)
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-5) and torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('cfg_name,batch', [('ssd_300_vgg16_voc', 32), ('ssd_512_vgg16_coco', 16), ('retina_rn50_500_coco', 8),
                                            ('retina_rn50_500_coco', 32), ('m2det_512_vgg16_coco', 16)])
@pytest.mark.parametrize('density', ['dense', 'sampled'])
def test_heads_adjoint_identity_at_baseline_size(cfg_name, batch, density):
    """Size-independent property at BASELINE.json's full sizes (no CPU reference needed): the heads are bilinear in (x, w), so for any
    upstream gradient g   <g, y - bias> = <dx, x> = <dw, w>   and   <g, bias broadcast> = <db, b>.
    'dense' runs the dense backward kernels, 'sampled' a hard-negative-mining-like gradient (4 % of the anchors), i.e. the
    anchor-granular sparse form."""
    from single_shot_detection_amd import synthetic as syn
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(11)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    with torch.no_grad():
        for p in heads.parameters():
            p.copy_(torch.randn_like(p) * (0.05 if p.dim() > 1 else 0.5))
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for cin, h, _ in levels]
    scores, locs = multi_level_heads(xs, xs, heads)
    A = scores.shape[1] // C
    assert A == syn.num_anchors(cfg) and locs.shape[1] == 4 * A
    gs, gl = torch.randn_like(scores), torch.randn_like(locs)
    if density == 'sampled':
        keep = (torch.rand((batch, A, 1), device='cuda') < 0.04).float()
        gs = (gs.view(batch, A, C) * keep).view(batch, -1)
        gl = (gl.view(batch, A, 4) * keep).view(batch, -1)
    torch.autograd.backward([scores, locs], [gs, gl])
    lhs = (scores.detach().double() * gs.double()).sum() + (locs.detach().double() * gl.double()).sum()
    bias_part = sum((p.grad.double() * p.detach().double()).sum() for n, p in heads.named_parameters() if n.endswith('bias'))
    via_x = sum((x.grad.double() * x.detach().double()).sum() for x in xs)
    via_w = sum((p.grad.double() * p.detach().double()).sum() for n, p in heads.named_parameters() if n.endswith('weight'))
    scale = float((scores.detach().double().abs() * gs.double().abs()).sum() + (locs.detach().double().abs() * gl.double().abs()).sum())
    tol = 2e-6 * scale   # fp32 products summed in fp32 inside the GEMMs: relative to the sum of magnitudes
    assert abs(float(lhs - bias_part - via_x)) <= tol, (float(lhs), float(bias_part), float(via_x), tol)
    assert abs(float(lhs - bias_part - via_w)) <= tol, (float(lhs), float(bias_part), float(via_w), tol)


@pytest.mark.parametrize('cfg_name,batch', [('ssd_300_vgg16_voc', 4), ('ssd_512_vgg16_coco', 2), ('retina_rn50_500_coco', 2), ('m2det_512_vgg16_coco', 2)])
def test_fast_mode_bf16x3_forward_vs_fp32(cfg_name, batch):
    """Opt-in fast mode (heads.set_fast_mode('bf16x3'): split-bf16 operands, three cross terms, fp32 accumulate on v_mfma_f32_32x32x16_bf16;
    the reference's analogue is apex AMP O1, bf/training/env.py:87-95) against the exact fp32 path AND torch's fp32 CPU convolution on the
    first level: every logit / loc within 1e-4 of the tensor's scale (a product is wrong by ~2^-16 relative, random in sign over K = 9 Cin
    terms), and the multibox loss of the same batch within north_star's 1e-4."""
    import functools
    from single_shot_detection_amd import synthetic as syn
    from single_shot_detection_amd.detection import anchor_generators, sampler
    from single_shot_detection_amd.detection.box_coder import BoxCoder
    from single_shot_detection_amd.detection.losses.multibox_loss import MultiboxLoss
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    from single_shot_detection_amd.detection.target_assigner import TargetAssigner
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(3)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C,
                                       score_head_bias_init=(-4.6 if cfg['loss'] != 'ce_hnm' else 0.0)).cuda()
    with torch.no_grad():
        for p in heads.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * 0.03)
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last) for cin, h, _ in levels]
    with torch.no_grad():
        s32, l32 = multi_level_heads(xs, xs, heads)
        prev = heads_mod.set_fast_mode('bf16x3')
        try:
            sf, lf = multi_level_heads(xs, xs, heads)
        finally:
            heads_mod.set_fast_mode(prev)
    for a, b in ((sf, s32), (lf, l32)):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale, (float((a - b).abs().max()), scale)
    assert float((sf - s32).abs().max()) > 0.0   # (it IS another arithmetic)
    # first level against torch's CPU convolution
    h0 = heads[0]
    ref = F.conv2d(xs[0].cpu(), h0['score'].weight.detach().cpu().contiguous(), h0['score'].bias.detach().cpu(), padding=1).permute(0, 2, 3, 1).reshape(batch, -1)
    n0 = ref.shape[1]
    assert float((sf[:, :n0].cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # the loss the two sets of predictions give on the same targets
    p = dict(cfg['anchor'])
    gens = getattr(anchor_generators, p.pop('type')).build_anchor_generators(**p)
    img = torch.empty((1, 3, cfg['size'], cfg['size']), device='cuda')
    anchors = torch.cat([g.generate(img, (h, h)).reshape(-1) for g, (_, h, _) in zip(gens, levels)]).view(-1, 4)
    gt = [torch.from_numpy(g).cuda() for g in syn.make_ground_truth(batch, cfg['size'], C, seed=1, background=cfg['score_converter'] == 'SOFTMAX')]
    if cfg['loss'] == 'ce_hnm':
        smp = functools.partial(sampler.hard_negative_mining, negative_per_positive_ratio=3, min_negative_per_image=5)
        cl = {'name': 'CrossEntropyLoss'}
    else:
        smp, cl = sampler.naive_sampler, {'name': 'SigmoidFocalLoss', 'gamma': 2.0, 'alpha': 0.25}
    crit = MultiboxLoss(sampler=smp, box_coder=BoxCoder(10.0, 5.0), classification_loss=cl, localization_loss={'name': 'SmoothL1Loss'})
    assigner = TargetAssigner(cfg['matched'], cfg['unmatched'])
    losses = []
    for sc, lo in ((s32, l32), (sf, lf)):
        target = assigner.encode_ground_truth(gt, anchors)
        losses.append(float(crit((sc, lo), anchors, target)[0]))
    assert abs(losses[0] - losses[1]) <= 1e-4 * max(1.0, abs(losses[0])), losses


def test_fast_mode_refuses_channel_counts_it_cannot_take():
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    heads = detector_builder.get_heads([24], [3], 7).cuda()
    x = torch.randn((1, 24, 5, 5), device='cuda')
    prev = heads_mod.set_fast_mode('bf16x3')
    try:
        with pytest.raises(ValueError):
            multi_level_heads([x], [x], heads)
    finally:
        heads_mod.set_fast_mode(prev)
    with pytest.raises(ValueError):
        heads_mod.set_fast_mode('fp8')


@pytest.mark.parametrize('cin,cout,k,stride,pad,relu,sizes', [
    (256, 256, 3, 1, 1, True, (32, 16, 8)),     # a RetinaNet tower layer: one weight tensor over several pyramid levels
    (1024, 256, 1, 1, 0, True, (19,)),          # the SSD tail's 1 x 1
    (256, 512, 3, 2, 1, False, (19,)),          # ... and its strided 3 x 3
    (64, 40, 3, 1, 1, False, (7,)),             # Cout not a multiple of 32
])
def test_fast_mode_bf16x3_generic_convolutions_vs_fp32(cin, cout, k, stride, pad, relu, sizes, monkeypatch):
    """ops.conv2d in the opt-in split-bf16 mode (ssdk_conv2d_fwd_fast) against the fp32 kernel and torch's CPU convolution: within 1e-4
    of the output's scale; the backward pass (fp32, unchanged) gives the same gradients for the same upstream gradient."""
    from single_shot_detection_amd import ops
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    monkeypatch.setenv('SSDK_FAST_MIN_FLOPS', '0')   # (these small shapes too: by default launches below ~1 GFLOP stay fp32)
    torch.manual_seed(5)
    w = (torch.randn((cout, cin, k, k), device='cuda') * 0.03).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = (torch.randn((cout,), device='cuda') * 0.1).requires_grad_(True)
    xs = [torch.randn((2, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for h in sizes]
    y32 = ops.conv2d(xs, w, b, stride, pad, relu=relu)
    prev = heads_mod.set_fast_mode('bf16x3')
    try:
        yf = ops.conv2d(xs, w, b, stride, pad, relu=relu)
    finally:
        heads_mod.set_fast_mode(prev)
    differs = False
    for a, r, x in zip(yf, y32, xs):
        a, r = a.detach(), r.detach()
        scale = float(r.abs().max())
        assert float((a - r).abs().max()) <= 1e-4 * scale, (float((a - r).abs().max()), scale)
        differs = differs or float((a - r).abs().max()) > 0.0
        ref = F.conv2d(x.detach().cpu(), w.detach().cpu().contiguous(), b.detach().cpu(), stride=stride, padding=pad)
        ref = F.relu(ref) if relu else ref
        assert float((a.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert differs   # (it IS another arithmetic)
    if relu:   # (an output within rounding of zero may be masked in one arithmetic and not in the other: one such element is |g x| of a
        return  # weight gradient -- seen: 6.3 on a scale of 180 -- so the fp32 backward is compared where there is no mask)
    gs = [torch.randn_like(r) for r in y32]
    g32 = torch.autograd.grad([r for r in y32], [w, b] + xs, gs)
    gf = torch.autograd.grad([a for a in yf], [w, b] + xs, gs)
    for a, r in zip(gf, g32):
        assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-6, (float((a - r).abs().max()), float(r.abs().max()))


def test_fast_mode_conv_batch_norm_block_trains_like_fp32(monkeypatch):
    monkeypatch.setenv('SSDK_FAST_MIN_FLOPS', '0')
    _fast_mode_conv_batch_norm_block_trains_like_fp32()


def _fast_mode_conv_batch_norm_block_trains_like_fp32():
    """Conv2dBn (conv -> BatchNorm statistics -> apply -> ReLU) with the convolution in the split-bf16 mode: output, running statistics
    and num_batches_tracked of a train() step next to the fp32 block's."""
    import copy
    from single_shot_detection_amd.bf.modules.conv import Conv2dBn
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    torch.manual_seed(7)
    blk = Conv2dBn(128, 64, kernel_size=3, padding=1, bias=False).cuda().to(memory_format=torch.channels_last).train()
    blk_f = copy.deepcopy(blk)
    x = torch.randn((4, 128, 10, 10), device='cuda').contiguous(memory_format=torch.channels_last)
    y32 = blk(x)
    prev = heads_mod.set_fast_mode('bf16x3')
    try:
        yf = blk_f(x)
    finally:
        heads_mod.set_fast_mode(prev)
    assert float((yf.detach() - y32.detach()).abs().max()) <= 1e-4 * float(y32.detach().abs().max())
    bn32 = [m for m in blk.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
    bnf = [m for m in blk_f.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
    assert int(bnf.num_batches_tracked) == int(bn32.num_batches_tracked) == 1
    assert float((bnf.running_mean - bn32.running_mean).abs().max()) <= 1e-5
    assert float((bnf.running_var - bn32.running_var).abs().max()) <= 1e-5 * float(bn32.running_var.abs().max())


def _cpu_heads(xs_cpu, heads_cpu):
    """detection/detector.py:50-66 on torch's fp32 CPU convolution for a (sub-)batch: (scores, locs) rows."""
    B = xs_cpu[0].shape[0]
    sc, lo = [], []
    for x, (ws, bs, wl, bl) in zip(xs_cpu, heads_cpu):
        sc.append(F.conv2d(x, ws, bs, padding=1).permute(0, 2, 3, 1).reshape(B, -1))
        lo.append(F.conv2d(x, wl, bl, padding=1).permute(0, 2, 3, 1).reshape(B, -1))
    return torch.cat(sc, 1), torch.cat(lo, 1)


@pytest.mark.parametrize('cfg_name,batch', [('ssd_300_vgg16_voc', 32), ('ssd_300_vgg16_voc', 64), ('ssd_512_vgg16_coco', 16),
                                            ('m2det_512_vgg16_coco', 16)])
def test_headline_launch_elementwise_vs_torch_cpu(cfg_name, batch):
    """The EXACT launches bench.py times (BASELINE.json's batch sizes: the stream-K range cuts, `parts` and the XCD remapping of
    igemm_streamk_kernel depend on the batch), checked element by element: images are independent, so torch's fp32 CPU convolution of
    images {0, B/2, B-1} pins those rows of scores / locs (2e-6 * sqrt(K), K = 9 Cin), and -- for a hard-negative-mining-like sampled
    gradient over the WHOLE batch (the anchor-granular backward the step runs) -- their rows of dx.  dw / db add over images (the
    heads are bilinear): a second backward with the gradient confined to the three images is compared with the CPU's sum over them."""
    from single_shot_detection_amd import synthetic as syn
    from single_shot_detection_amd import _lib
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(29)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    with torch.no_grad():
        for p in heads.parameters():
            p.copy_(torch.randn_like(p) * (0.02 if p.dim() > 1 else 0.1))
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for cin, h, _ in levels]
    picks = [0, batch // 2, batch - 1]
    scores, locs = multi_level_heads(xs, xs, heads)
    A = scores.shape[1] // C
    # the reference on the three images
    xs_cpu = [x.detach()[picks].cpu().contiguous().requires_grad_(True) for x in xs]
    heads_cpu = [tuple(t.detach().cpu().contiguous().requires_grad_(True) for t in (h['score'].weight, h['score'].bias, h['loc'].weight, h['loc'].bias))
                 for h in heads]
    sc_ref, lo_ref = _cpu_heads(xs_cpu, heads_cpu)
    tol = 2e-6 * np.sqrt(9 * max(l[0] for l in levels))
    np.testing.assert_allclose(scores.detach()[picks].cpu().numpy(), sc_ref.detach().numpy(), rtol=1e-5, atol=tol)
    np.testing.assert_allclose(locs.detach()[picks].cpu().numpy(), lo_ref.detach().numpy(), rtol=1e-5, atol=tol)
    # sampled gradient, every image: ~4 % of the anchors carry one (all their classes + their box)
    keep = (torch.rand((batch, A, 1), device='cuda') < 0.04).float()
    gs = (torch.randn_like(scores).view(batch, A, C) * keep).view(batch, -1)
    gl = (torch.randn_like(locs).view(batch, A, 4) * keep).view(batch, -1)
    torch.autograd.backward([scores, locs], [gs, gl], retain_graph=True)
    torch.autograd.backward([sc_ref, lo_ref], [gs[picks].cpu(), gl[picks].cpu()])
    for i in range(len(levels)):
        ref = xs_cpu[i].grad.numpy()
        scale = float(np.abs(ref).max())
        np.testing.assert_allclose(xs[i].grad[picks].cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * scale + 1e-6, err_msg=f'dx level {i}')
    # dw / db: the same gradient confined to the three images
    for p in heads.parameters():
        p.grad = None
    only = torch.zeros((batch, 1), device='cuda')
    only[picks] = 1.0
    torch.autograd.backward([scores, locs], [gs * only, gl * only])
    for i, (h, hc) in enumerate(zip(heads, heads_cpu)):
        for name, t, r in (('score.weight', h['score'].weight, hc[0]), ('score.bias', h['score'].bias, hc[1]),
                           ('loc.weight', h['loc'].weight, hc[2]), ('loc.bias', h['loc'].bias, hc[3])):
            ref = r.grad.numpy()
            scale = float(np.abs(ref).max()) + 1e-12
            np.testing.assert_allclose(t.grad.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * scale + 1e-6, err_msg=f'level {i} {name}')
    assert _lib.streamk_timeouts() == 0


@pytest.mark.parametrize('cfg_name,batch,side,main_wgs,one_launch,ordered', [
    ('ssd_300_vgg16_voc', 32, True, 448, False, False), ('ssd_300_vgg16_voc', 32, False, 0, False, False),
    ('ssd_300_vgg16_voc', 4, True, 0, False, True), ('ssd_512_vgg16_coco', 16, True, 480, False, False),
    ('ssd_300_vgg16_voc', 32, True, 0, True, False), ('ssd_300_vgg16_voc', 32, True, 0, True, True), ('ssd_512_vgg16_coco', 16, False, 0, True, False)])
def test_split_heads_match_the_single_launch(cfg_name, batch, side, main_wgs, one_launch, ordered):
    """multi_level_heads_split (two autograd nodes: the backbone-tap levels on the current stream, the tail's levels on a second one, a
    join that hands the loss' gradient rows to both) against multi_level_heads on the same inputs: same scores / locs rows and the
    same gradients up to the order of the fp32 sums (the stream-K ranges of the two part launches are cut elsewhere)."""
    from single_shot_detection_amd import synthetic as syn
    from single_shot_detection_amd import _lib
    from single_shot_detection_amd.detection.modules.heads import multi_level_heads_split
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(31)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    with torch.no_grad():
        for p in heads.parameters():
            p.copy_(torch.randn_like(p) * (0.02 if p.dim() > 1 else 0.1))
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for cin, h, _ in levels]
    s0, l0 = multi_level_heads(xs, xs, heads)
    A = s0.shape[1] // C
    keep = (torch.rand((batch, A, 1), device='cuda') < 0.04).float()
    gs = (torch.randn_like(s0).view(batch, A, C) * keep).view(batch, -1)
    gl = (torch.randn_like(l0).view(batch, A, 4) * keep).view(batch, -1)
    params = list(heads.parameters())
    g0 = torch.autograd.grad([s0, l0], xs + params, [gs, gl])
    stream = torch.cuda.Stream(priority=-1) if side else None
    # the "tail" of this test: a scaled copy made on the side stream, so that the levels behind the split are produced there
    xs2 = [x.detach().clone().requires_grad_(True) for x in xs]

    def run_tail():
        return [x * 1.0 for x in xs2[2:]]
    s1, l1, srcs = multi_level_heads_split(xs2[:2], heads, 2, run_tail, side_stream=stream, main_workgroups=main_wgs, one_launch=one_launch,
                                           ordered_backward=ordered)
    if one_launch:   # (the same grouped launch as multi_level_heads: the same bits)
        assert torch.equal(s1.detach(), s0.detach()) and torch.equal(l1.detach(), l0.detach())
    assert len(srcs) == len(levels)
    tol = 2e-6 * np.sqrt(9 * max(l[0] for l in levels))
    np.testing.assert_allclose(s1.detach().cpu().numpy(), s0.detach().cpu().numpy(), rtol=1e-5, atol=tol)
    np.testing.assert_allclose(l1.detach().cpu().numpy(), l0.detach().cpu().numpy(), rtol=1e-5, atol=tol)
    g1 = torch.autograd.grad([s1, l1], xs2 + params, [gs, gl])
    torch.cuda.synchronize()
    for a, b in zip(g1, g0):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-6, (tuple(b.shape), float((a - b).abs().max()), scale)
    assert _lib.streamk_timeouts() == 0


def test_streamk_timeout_is_loud_in_graph_replays(tmp_path):
    """A stream-K partner that never arrives (fault injection, tests/streamk_fault_worker.py in a child process: the poison is sticky per
    process) -- in a HIP-graph REPLAY, which never passes through ssdk_heads_fwd's host-side check: the losing replay has NaN in the
    owner's tile and only there, the sticky word is set, GraphedCallable refuses the next call, a raw replay afterwards stores NaN in
    every tile (not the previous replay's stale partial sums), and the eager entry point fails."""
    import json
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'streamk_fault_worker.py')
    out = os.path.join(str(tmp_path), 'fault.json')
    r = subprocess.run([sys.executable, worker, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, SSDK_ENABLE_FAULT_INJECTION='1'))
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res['healthy_finite'] and not res['healthy_poisoned']
    assert res['nan_in_faulted_replay'] > 0 and res['untouched_rows_equal'], res
    assert res['poisoned_after_fault'] and res['timeouts'] >= 1, res
    assert res['graphed_callable_raises'] and res['all_nan_in_later_replay'] and res['eager_raises'], res
    # ssdk_streamk_reset: the process recovers without a restart (advisor, round 4)
    assert not res['poisoned_after_reset'] and res['timeouts_after_reset'] == 0, res
    assert res['replay_after_reset_equals_healthy'] and res['eager_after_reset_equals_healthy'], res
    # ... and the fault hook itself is refused unless the process asked for fault injection
    from single_shot_detection_amd import _lib
    if not os.environ.get('SSDK_ENABLE_FAULT_INJECTION'):
        assert _lib.lib().ssdk_debug_streamk_fault(-1, 0) == -3


@pytest.mark.parametrize('levels,C,B,density', [(SSD300_LEVELS, 81, 2, 0.05), (SSD300_LEVELS, 21, 2, 1.0), ([(24, 7, 3), (40, 5, 5)], 7, 3, 0.4),
                                                ([(256, 8, 9)], 80, 1, 0.05)])
def test_deterministic_mode_heads_vs_torch_cpu_conv(levels, C, B, density, monkeypatch):
    """The heads under ops.deterministic(): the dense data gradient, K-split weight-gradient copies added in split order and two-stage
    bias column sums against torch's CPU convolution (same bounds as the default forms), for sampled and dense upstream gradients."""
    from single_shot_detection_amd import ops
    with ops.deterministic():
        _heads_vs_torch_cpu_conv(levels, C, B, density, None, monkeypatch)


@pytest.mark.parametrize('cfg_name,batch', [('retina_rn50_500_coco', 2), ('ssd_300_vgg16_voc', 2), ('m2det_512_vgg16_coco', 1)])
def test_fast_mode_bf16x3_heads_backward_vs_fp32(cfg_name, batch):
    """heads.fast_mode('bf16x3') in the backward pass (ssdk_heads_bwd_fast): the DENSE data gradient -- what a focal-loss step takes on
    every level -- as the forward convolution of the packed gradient rows with the mirrored kernel on the split-bf16 GEMM, against the
    fp32 kernels on the same dense upstream gradient: dx within 1e-4 of its scale on every level (and not identical: it IS another
    arithmetic); the weight gradients take the same mode (igemm_wgrad_bf16x3_kernel: both operands split in registers), bias gradients
    stay fp32 column sums."""
    from single_shot_detection_amd import synthetic as syn
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(41)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    with torch.no_grad():
        for p in heads.parameters():
            p.copy_(torch.randn_like(p) * (0.03 if p.dim() > 1 else 0.1))
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for cin, h, _ in levels]
    params = list(heads.parameters())
    scores, locs = multi_level_heads(xs, xs, heads)
    gs, gl = torch.randn_like(scores), torch.randn_like(locs)     # dense: every anchor carries a gradient
    g32 = torch.autograd.grad([scores, locs], xs + params, [gs, gl], retain_graph=True)
    with heads_mod.fast_mode('bf16x3'):
        gf = torch.autograd.grad([scores, locs], xs + params, [gs, gl])
    differs = False
    for i in range(len(xs)):
        scale = float(g32[i].abs().max())
        err = float((gf[i] - g32[i]).abs().max())
        assert err <= 1e-4 * scale, (i, err, scale)
        differs = differs or err > 0.0
    assert differs
    for a, r in zip(gf[len(xs):], g32[len(xs):]):   # weight gradients: the split-bf16 form of the same K-split GEMM (biases: fp32 column sums)
        assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-7


def test_fast_mode_bf16x3_training_steps_track_fp32():
    """Four training steps of the RetinaNet hot path (towers + heads, focal loss, dense gradients) with forward AND data gradients in the
    split-bf16 mode against the fp32 step from the same state: the loss of every step within 1e-4 (relative), the heads' gradients
    of the first step within 1e-3 of each tensor's scale (both sides in deterministic mode: no atomics noise).  The towers' own gradients
    are not compared tensor by tensor: conv -> ReLU -> BatchNorm over as few as 32 rows turns a 1e-5 perturbation of the activations into
    flipped ReLU masks, i.e. discrete changes of small gradients (seen: 2.5e-5 on a tensor whose largest gradient is 1.3e-4); layer by
    layer the arithmetic is held to 1e-4 by test_fast_mode_bf16x3_generic_convolutions_vs_fp32 / ..._heads_backward_vs_fp32 (the reference's analogue, apex AMP O1, is far coarser)."""
    import bench
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    from test_end_to_end_gpu import _copy_state
    from single_shot_detection_amd import ops
    dev = torch.device('cuda:0')
    with ops.deterministic():   # (no fp32 atomics on either side: what differs is the arithmetic of the GEMMs alone)
        a, b = bench.HotPath('retina_rn50_500_coco', 2, dev), bench.HotPath('retina_rn50_500_coco', 2, dev)
        a.train_step()
        b.train_step()
        _copy_state(a, b)
        for k in range(4):
            la = float(a.train_step().detach())
            with heads_mod.fast_mode('bf16x3'):
                lb = float(b.train_step().detach())
            assert abs(la - lb) <= 1e-4 * abs(la), (k, la, lb)
            if k == 0:   # one step from the same state: the heads' gradients (the bulk of the parameters) within 1e-3 of each tensor's scale
                for i, (p, q) in enumerate(zip(a.head_params, b.head_params)):
                    scale = float(p.grad.abs().max()) + 1e-12
                    assert float((p.grad - q.grad).abs().max()) <= 1e-3 * scale, (i, tuple(p.shape), float((p.grad - q.grad).abs().max()), scale)


@pytest.mark.parametrize('cfg_name,batch', [('ssd_300_vgg16_voc', 8), ('ssd_mb2_voc', 2), ('ssd_512_vgg16_coco', 3)])
def test_heads_backward_with_a_row_mask_equals_the_full_scan(cfg_name, batch):
    """ssdk_heads_bwd_ex with the gradient producer's row mask (0 = that anchor's rows are zeros): the pack pass skips the pixels whose
    anchors are all unmarked instead of scanning dscores -- same dx / dw / db as the full scan (up to the order of the atomics), for the
    exact mask and for a superset of it (extra anchors marked, as a positive without a classification term would be)."""
    from single_shot_detection_amd import synthetic as syn
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(43)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for cin, h, _ in levels]
    params = list(heads.parameters())
    scores, locs = multi_level_heads(xs, xs, heads)
    A = scores.shape[1] // C
    keep = torch.rand((batch, A), device='cuda') < 0.04
    gs = (torch.randn_like(scores).view(batch, A, C) * keep[..., None]).view(batch, -1)
    gl = (torch.randn_like(locs).view(batch, A, 4) * keep[..., None]).view(batch, -1)
    ref = torch.autograd.grad([scores, locs], xs + params, [gs, gl], retain_graph=True)
    for extra in (0.0, 0.02):
        mask = (keep | (torch.rand((batch, A), device='cuda') < extra)).to(torch.uint8).contiguous()
        taken = heads_mod.row_hints_taken

        class Producer(torch.autograd.Function):   # a gradient producer that announces its mask like MultiboxLoss's backward does
            @staticmethod
            def forward(ctx, s, l):
                return (s * 0).sum() + (l * 0).sum()

            @staticmethod
            def backward(ctx, g):
                ds, dl = gs.clone(), gl.clone()
                heads_mod.set_row_hint(ds, dl, mask)
                return ds, dl
        got = torch.autograd.grad(Producer.apply(scores, locs), xs + params, retain_graph=True)
        assert heads_mod.row_hints_taken == taken + 1
        for a, r in zip(got, ref):
            assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-7
    # a hint for OTHER tensors (autograd added a second term on the way: the heads receive a new tensor) is not taken
    taken = heads_mod.row_hints_taken

    class Producer2(torch.autograd.Function):
        @staticmethod
        def forward(ctx, s, l):
            return (s * 0).sum() + (l * 0).sum()

        @staticmethod
        def backward(ctx, g):
            ds, dl = gs.clone(), gl.clone()
            heads_mod.set_row_hint(ds, dl, torch.zeros((batch, A), dtype=torch.uint8, device='cuda'))   # (an all-zero mask: taking it would lose everything)
            return ds, dl
    total = Producer2.apply(scores, locs) + 1e-3 * scores.sum()
    got = torch.autograd.grad(total, xs + params, retain_graph=True)
    assert heads_mod.row_hints_taken == taken
    ref2 = torch.autograd.grad([scores, locs], xs + params, [gs + 1e-3, gl])
    for a, r in zip(got, ref2):
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-7


def _ordered_layout(levels_meta, heads, xs, batch, level):
    """ssdk_debug_heads_bwd_layout for one level of a heads call -> (offsets, workspace bytes as numpy)."""
    import ctypes as C
    from single_shot_detection_amd import _lib
    from single_shot_detection_amd.detection.modules import heads as H
    lvs, s_off, l_off = [], 0, 0
    for (cin, h, nb), head, x in zip(levels_meta, heads, xs):
        ws, wl = H.weight_khwc(head['score'].weight.detach()), H.weight_khwc(head['loc'].weight.detach())
        lvs.append(dict(x=H.to_nhwc(x.detach()), H=h, W=h, cin=cin, ws=ws, bs=None, wl=wl, bl=None, ns=ws.shape[0], nl=wl.shape[0], s_off=s_off, l_off=l_off))
        s_off += h * h * ws.shape[0]
        l_off += h * h * wl.shape[0]
    arr = H._level_array(lvs)
    out = (C.c_ulonglong * 8)()
    rc = _lib.lib().ssdk_debug_heads_bwd_layout(arr, len(lvs), batch, level, out)
    assert rc == 0, rc
    return [int(v) for v in out], _lib.scratch(0, xs[0].device, 'heads_bwd').cpu().numpy()


@pytest.mark.parametrize('levels,C,B,density', [([(64, 6, 4), (32, 3, 6)], 21, 2, 0.1), ([(512, 18, 6)], 81, 2, 1.0), ([(96, 19, 4), (1280, 10, 6)], 21, 2, 0.05)])
def test_ordered_backward_intermediates(levels, C, B, density, monkeypatch):
    """The intermediates of the ordered anchor-row backward, through ssdk_debug_heads_bwd_layout: rows numbered in PIXEL order per anchor
    type (whatever order the workgroups ran in), the inverted index aidx = the rows' T positions and -1 elsewhere, the row matrix = the
    anchors' C + 4 gradient values, and T[row] = row . W_type against numpy in fp64 -- the contribution rows anchor_dx_kernel sums
    (detection/detector.py:50-66 backward; detection/losses/multibox_loss.py:60-90 decides which anchors carry a gradient)."""
    monkeypatch.delenv('SSDK_HEADS_BWD_MODE', raising=False)
    rng = np.random.default_rng(5)
    weights, xs_np = {}, []
    for i, (cin, h, nb) in enumerate(levels):
        xs_np.append(rng.standard_normal((B, cin, h, h), dtype=np.float32))
        for k, nout in (('score', nb * C), ('loc', nb * 4)):
            weights[(k, i)] = (rng.standard_normal((nout, cin, 3, 3), dtype=np.float32) * np.float32(0.05), np.zeros((nout,), np.float32))
    heads = build_heads(levels, C, weights)
    xs = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs_np]
    scores, locs = multi_level_heads(xs, xs, heads)
    A = scores.shape[1] // C
    keep = rng.random((B, A)) < density
    gs = (rng.standard_normal((B, A, C), dtype=np.float32) * keep[..., None]).astype(np.float32)
    gl = (rng.standard_normal((B, A, 4), dtype=np.float32) * keep[..., None]).astype(np.float32)
    torch.autograd.grad([scores, locs], xs, [torch.from_numpy(gs).view(B, -1).cuda(), torch.from_numpy(gl).view(B, -1).cuda()])
    torch.cuda.synchronize()
    a_off = 0
    for li, (cin, h, nb) in enumerate(levels):
        off, wsb = _ordered_layout(levels, heads, xs, B, li)
        M, K9, Jpad = B * h * h, 9 * cin, (C + 4 + 31) // 32 * 32
        i32 = lambda o, n: wsb[o:o + 4 * n].view(np.int32)
        f32 = lambda o, n: wsb[o:o + 4 * n].view(np.float32)
        counts, plan, mode = i32(off[4], 16), i32(off[5], 34), int(i32(off[6], 1)[0])
        g_s = gs[:, a_off:a_off + h * h * nb].reshape(B * h * h, nb, C)
        g_l = gl[:, a_off:a_off + h * h * nb].reshape(B * h * h, nb, 4)
        marked = keep[:, a_off:a_off + h * h * nb].reshape(B * h * h, nb)
        a_off += h * h * nb
        assert mode == 2, (li, mode)
        ga = f32(off[0], nb * M * Jpad).reshape(nb, M, Jpad)
        apix, aidx = i32(off[3], nb * M).reshape(nb, M), i32(off[2], nb * M).reshape(nb, M)
        rows = int(plan[nb])
        assert rows == int(marked.sum()) and rows <= off[7]
        T = f32(off[1], rows * K9).reshape(rows, K9)
        wsk = weights[('score', li)][0].transpose(0, 2, 3, 1).reshape(nb, C, K9)      # [type][j][tap * Cin + c]
        wlk = weights[('loc', li)][0].transpose(0, 2, 3, 1).reshape(nb, 4, K9)
        for k in range(nb):
            n = int(counts[k])
            pix = np.nonzero(marked[:, k])[0]
            assert n == len(pix) and np.array_equal(apix[k, :n], pix), 'rows are the marked pixels of the type, in pixel order'
            assert np.array_equal(aidx[k, pix], plan[k] + np.arange(n)) and int((aidx[k] >= 0).sum()) == n and np.all(aidx[k][~marked[:, k]] == -1)
            if not n:
                continue
            ref_rows = np.concatenate([g_s[pix, k], g_l[pix, k]], 1)
            assert np.array_equal(ga[k, :n, :C + 4], ref_rows) and np.all(ga[k, :n, C + 4:] == 0)
            Wk = np.concatenate([wsk[k], wlk[k]], 0)
            Tref = ref_rows.astype(np.float64) @ Wk.astype(np.float64)
            np.testing.assert_allclose(T[plan[k]:plan[k] + n], Tref, rtol=1e-5, atol=2e-6 * np.sqrt(C + 4) * float(np.abs(Tref).max()))


@pytest.mark.parametrize('cfg_name,batch', [('ssd_300_vgg16_voc', 8), ('ssd_mb2_voc', 2)])
def test_heads_backward_is_bitwise_reproducible_by_default(cfg_name, batch):
    """The ordered pipeline sums in an order fixed by the launch -- rows in pixel order, taps and anchor types in order, K splits added
    in split order, bias chunks in chunk order -- so the heads' backward gives the same bits on every run WITHOUT ops.deterministic()
    (the reference runs cudnn.deterministic = True, bf/training/env.py:74-76), with the row mask derived from the gradient or given."""
    from single_shot_detection_amd import synthetic as syn
    cfg = syn.CONFIGS[cfg_name]
    levels, C = cfg['levels'], cfg['num_classes']
    torch.manual_seed(47)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    xs = [torch.randn((batch, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for cin, h, _ in levels]
    params = list(heads.parameters())
    scores, locs = multi_level_heads(xs, xs, heads)
    A = scores.shape[1] // C
    keep = torch.rand((batch, A), device='cuda') < 0.04
    gs = (torch.randn_like(scores).view(batch, A, C) * keep[..., None]).view(batch, -1)
    gl = (torch.randn_like(locs).view(batch, A, 4) * keep[..., None]).view(batch, -1)
    first = [g.clone() for g in torch.autograd.grad([scores, locs], xs + params, [gs, gl], retain_graph=True)]
    for _ in range(3):
        again = torch.autograd.grad([scores, locs], xs + params, [gs, gl], retain_graph=True)
        for a, r in zip(again, first):
            assert torch.equal(a, r)


@pytest.mark.parametrize('levels,C,B,density,det', [
    ([(64, 16, 9), (64, 8, 9), (64, 4, 9)], 80, 2, 0.05, False),     # RetinaNet-like: 9 anchor types, 80 sigmoid classes, few anchors with a gradient
    ([(64, 16, 9), (64, 8, 9), (64, 4, 9)], 80, 2, 0.05, True),      # ... the same kernels in deterministic mode
    ([(64, 16, 9), (64, 8, 9)], 80, 2, 1.0, False),                   # every anchor: the rows do not fit T, the dense form on the device
    ([(96, 9, 3), (32, 5, 6)], 7, 3, 0.3, False),                     # other widths (C = 7 -> 32 columns; loc rows of 4 -> 32)
])
def test_single_head_towers_backward_vs_torch_cpu_conv(levels, C, B, density, det, monkeypatch):
    """multi_level_heads with DIFFERENT source maps for the score and the loc heads (SharedConvPredictor: two towers, detector.py:50-66):
    two single-head calls, each without the partner head that tells the library how many anchor types a pixel has -- the count travels in
    the level's locs_offset.  The ordered anchor-row backward (round 5) then covers them: data, weight and bias gradients of both towers'
    heads against torch's CPU convolution, for sparse (anchor rows) and dense upstream gradients, default and deterministic mode."""
    monkeypatch.delenv('SSDK_HEADS_BWD_MODE', raising=False)
    _single_heads_vs_torch_cpu_conv(levels, C, B, density, det, check_layout=True)


def _single_heads_vs_torch_cpu_conv(levels, C, B, density, det, seed=23, check_layout=False):
    from single_shot_detection_amd import ops
    rng = np.random.default_rng(seed)
    weights, xs_np, xl_np = {}, [], []
    for i, (cin, h, nb) in enumerate(levels):
        xs_np.append(rng.standard_normal((B, cin, h, h), dtype=np.float32))
        xl_np.append(rng.standard_normal((B, cin, h, h), dtype=np.float32))
        for k, nout in (('score', nb * C), ('loc', nb * 4)):
            weights[(k, i)] = (rng.standard_normal((nout, cin, 3, 3), dtype=np.float32) * np.float32(0.02),
                               rng.standard_normal((nout,), dtype=np.float32) * np.float32(0.1))
    xs_cpu = [torch.from_numpy(x).requires_grad_(True) for x in xs_np]
    xl_cpu = [torch.from_numpy(x).requires_grad_(True) for x in xl_np]
    ws_cpu = {k: (torch.from_numpy(w).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)) for k, (w, b) in weights.items()}
    s_ref = torch.cat([F.conv2d(x, *ws_cpu[('score', i)], padding=1).permute(0, 2, 3, 1).reshape(B, -1) for i, x in enumerate(xs_cpu)], 1)
    l_ref = torch.cat([F.conv2d(x, *ws_cpu[('loc', i)], padding=1).permute(0, 2, 3, 1).reshape(B, -1) for i, x in enumerate(xl_cpu)], 1)
    A = sum(h * h * nb for _, h, nb in levels)
    keep = torch.from_numpy((rng.random((B, A)) < density).astype(np.float32))        # anchors that carry a gradient
    gs = torch.from_numpy(rng.standard_normal((B, A, C), dtype=np.float32)) * keep[..., None]
    gl = torch.from_numpy(rng.standard_normal((B, A, 4), dtype=np.float32)) * keep[..., None]
    gs, gl = gs.reshape(B, -1), gl.reshape(B, -1)
    ((s_ref * gs).sum() + (l_ref * gl).sum()).backward()

    heads = build_heads(levels, C, weights)
    xs = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs_np]
    xl = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xl_np]
    with ops.deterministic(det):
        scores, locs = multi_level_heads(xs, xl, heads)
        tol = 2e-6 * np.sqrt(9 * max(l[0] for l in levels))
        np.testing.assert_allclose(scores.detach().cpu().numpy(), s_ref.detach().numpy(), rtol=1e-5, atol=tol)
        np.testing.assert_allclose(locs.detach().cpu().numpy(), l_ref.detach().numpy(), rtol=1e-5, atol=tol)
        ((scores * gs.cuda()).sum() + (locs * gl.cuda()).sum()).backward()
    for i in range(len(levels)):
        np.testing.assert_allclose(xs[i].grad.cpu().numpy(), xs_cpu[i].grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(xl[i].grad.cpu().numpy(), xl_cpu[i].grad.numpy(), rtol=1e-4, atol=1e-4)
        for k in ('score', 'loc'):
            wref, bref = ws_cpu[(k, i)]
            scale = float(wref.grad.abs().max()) + 1e-6
            np.testing.assert_allclose(heads[i][k].weight.grad.cpu().numpy(), wref.grad.numpy(), rtol=1e-4, atol=2e-5 * scale + 1e-5)
            np.testing.assert_allclose(heads[i][k].bias.grad.cpu().numpy(), bref.grad.numpy(), rtol=1e-4, atol=1e-4)
    if not check_layout:
        return
    # (the ordered pipeline took these calls: its layout query answers for single-head levels that carry the anchor-type count)
    from single_shot_detection_amd import _lib
    import ctypes
    arr = (_lib.HeadLevel * 1)()
    arr[0].h, arr[0].w, arr[0].cin, arr[0].n_score, arr[0].n_loc, arr[0].locs_offset = levels[0][1], levels[0][1], levels[0][0], levels[0][2] * C, 0, levels[0][2]
    arr[0].x, arr[0].w_score = xs[0].data_ptr(), heads[0]['score'].weight.data_ptr()
    out = (ctypes.c_ulonglong * 8)()
    assert _lib.lib().ssdk_debug_heads_bwd_layout(arr, 1, B, 0, out) == 0
    arr[0].locs_offset = 0
    assert _lib.lib().ssdk_debug_heads_bwd_layout(arr, 1, B, 0, out) == -3
