"""GPU parity: anchors (A1/A2) and IoU-match-encode (T1/T2/T3) through the C ABI vs the oracle and the
reference's golden vectors.  Bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from single_shot_detection_amd import synthetic as syn
from single_shot_detection_amd.detection import anchor_generators
from single_shot_detection_amd.detection.target_assigner import TargetAssigner
from conftest import CONFIG_NAMES, GOLDEN_BATCH, load_golden

pytestmark = pytest.mark.gpu


def bits(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def device_anchors(cfg, dev='cuda'):
    p = dict(cfg['anchor'])
    gens = getattr(anchor_generators, p.pop('type')).build_anchor_generators(**p)
    img = torch.empty((1, 3, cfg['size'], cfg['size']), device=dev)
    return torch.cat([g.generate(img, (h, h)).reshape(-1) for g, (_, h, _) in zip(gens, cfg['levels'])]).view(-1, 4)


@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_anchors_bit_exact_vs_reference(name):
    cfg = syn.CONFIGS[name]
    a = device_anchors(cfg)
    assert np.array_equal(bits(a), bits(load_golden(name)['anchors']))


def test_kats_on_gpu(kats):
    for tag, thr in (('kat1', (0.5, 0.5)), ('kat3', (0.5, 0.5)), ('kat5', (0.9, 0.3))):
        anchors = torch.from_numpy(kats['kat_anchors']).cuda()
        t, idx = TargetAssigner(*thr).encode_ground_truth([torch.from_numpy(kats[f'{tag}_gt'])], anchors, return_box_idx=True)
        assert np.array_equal(idx[0].cpu().numpy(), kats[f'{tag}_idx']), tag
        assert np.array_equal(bits(t), bits(kats[f'{tag}_target'])), tag
    anchors = torch.from_numpy(kats['kat5b_anchors']).cuda()
    t = TargetAssigner(0.5, 0.4).encode_ground_truth([torch.from_numpy(kats['kat5b_gt'])], anchors)
    assert np.array_equal(bits(t), bits(kats['kat5b_target']))
    anchors = torch.from_numpy(kats['kat_anchors']).cuda()
    gt_list = [torch.zeros((0, 6)), torch.tensor([[5., 5., 15., 15., 1., 1.]])]
    t = TargetAssigner(0.5, 0.5).encode_ground_truth(gt_list, anchors)
    assert np.array_equal(bits(t), bits(kats['kat7_target']))


@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_match_bit_exact_vs_reference_golden(name):
    cfg = syn.CONFIGS[name]
    g = load_golden(name)
    softmax = cfg['score_converter'] == 'SOFTMAX'
    anchors = torch.from_numpy(g['anchors']).cuda()
    ta = TargetAssigner(cfg['matched'], cfg['unmatched'])
    gt = syn.make_ground_truth(GOLDEN_BATCH[name], cfg['size'], cfg['num_classes'], seed=1, background=softmax)
    t, idx = ta.encode_ground_truth([torch.from_numpy(x) for x in gt], anchors, return_box_idx=True)
    assert np.array_equal(idx.cpu().numpy(), g['match_box_idx'].astype(np.int32))
    assert np.array_equal(bits(t), bits(g['match_target']))
    gt32 = syn.make_ground_truth(2, cfg['size'], cfg['num_classes'], seed=11, fixed_g=32, background=softmax)
    t, idx = ta.encode_ground_truth([torch.from_numpy(x) for x in gt32], anchors, return_box_idx=True)
    assert np.array_equal(idx.cpu().numpy(), g['match32_box_idx'].astype(np.int32))


@pytest.mark.parametrize('name,batch', [('ssd_300_vgg16_voc', 32), ('ssd_300_vgg16_voc', 64), ('ssd_512_vgg16_coco', 16),
                                        ('retina_rn50_500_coco', 32), ('m2det_512_vgg16_coco', 16)])
def test_match_full_size_vs_oracle(name, batch):
    """BASELINE.json sizes; oracle finishes these in well under a second."""
    cfg = syn.CONFIGS[name]
    softmax = cfg['score_converter'] == 'SOFTMAX'
    anchors_np = load_golden(name)['anchors']
    gt = syn.make_ground_truth(batch, cfg['size'], cfg['num_classes'], seed=4, background=softmax)
    gt[3] = np.zeros((0, 6), np.float32)                       # an empty image in the middle
    gt[5] = np.concatenate([gt[5], gt[5][:1]], axis=0)         # duplicated box: tie on every anchor
    gt.append(gt.pop(0))                                        # ragged order
    ref_t, ref_idx = oracle.encode_ground_truth(gt, anchors_np, cfg['matched'], cfg['unmatched'], return_box_idx=True)
    t, idx = TargetAssigner(cfg['matched'], cfg['unmatched']).encode_ground_truth(
        [torch.from_numpy(x) for x in gt], torch.from_numpy(anchors_np).cuda(), return_box_idx=True)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(bits(t), bits(ref_t))


def test_match_many_boxes_chunked():
    """G = 300 > the 128-box LDS chunk."""
    cfg = syn.CONFIGS['ssd_300_vgg16_voc']
    anchors_np = load_golden('ssd_300_vgg16_voc')['anchors']
    gt = syn.make_ground_truth(3, cfg['size'], cfg['num_classes'], seed=9, fixed_g=300)
    ref_t, ref_idx = oracle.encode_ground_truth(gt, anchors_np, 0.5, 0.4, return_box_idx=True)
    t, idx = TargetAssigner(0.5, 0.4).encode_ground_truth([torch.from_numpy(x) for x in gt],
                                                           torch.from_numpy(anchors_np).cuda(), return_box_idx=True)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(bits(t), bits(ref_t))


def test_invalid_arguments_raise():
    anchors = torch.zeros((8, 4), device='cuda')
    with pytest.raises(ValueError):
        TargetAssigner(0.4, 0.5).encode_ground_truth([torch.zeros((1, 6))], anchors)
    from single_shot_detection_amd import _lib
    with pytest.raises(_lib.SsdkError):
        TargetAssigner(0.5, 0.5).encode_ground_truth([torch.zeros((1, 6))], anchors.cpu())


def test_match_per_prediction_kats_and_oracle(kats):
    """matcher.match_per_prediction(weights, ..., force_match_for_each_target) on a given matrix: the reference's own outputs
    (kats.npz: with and without force-matching, ties, the ignore band) and the oracle on a large random matrix with ties and NaNs."""
    from single_shot_detection_amd.detection import matcher
    for key, mt, ut in (('kat1', 0.5, 0.5), ('kat5', 0.9, 0.3)):
        w = torch.from_numpy(kats[f'{key}_iou']).cuda()
        assert np.array_equal(matcher.match_per_prediction(w, mt, ut).cpu().numpy(), kats[f'{key}_idx'])
        assert np.array_equal(matcher.match_per_prediction(w, mt, ut, force_match_for_each_target=False).cpu().numpy(), kats[f'{key}_idx_nf'])
    w = torch.from_numpy(kats['kat4_w']).cuda()
    assert np.array_equal(matcher.match_per_prediction(w, 0.5, 0.4).cpu().numpy(), kats['kat4_idx'])
    assert np.array_equal(matcher.match_per_prediction(w, 0.5, 0.4, force_match_for_each_target=False).cpu().numpy(), kats['kat4_idx_nf'])
    assert np.array_equal(matcher.match_per_prediction(torch.from_numpy(kats['kat3_iou']).cuda(), 0.5).cpu().numpy(), kats['kat3_idx'])
    rng = np.random.default_rng(8)
    w = rng.random((300, 8108), dtype=np.float32)
    w[:, ::7] = np.round(w[:, ::7] * 8) / 8          # many exact ties down a column and along a row
    w[5] = w[4]                                       # two boxes with identical rows: the later one wins their common best anchor
    w[17, 100] = np.nan
    for force in (True, False):
        ref = oracle.match_per_prediction(w, 0.5, 0.4, force)
        got = matcher.match_per_prediction(torch.from_numpy(w).cuda(), 0.5, 0.4, force).cpu().numpy()
        assert np.array_equal(got, ref)
    with pytest.raises(ValueError):
        matcher.match_per_prediction(torch.zeros((0, 8), device='cuda'), 0.5)


def test_ssd_anchor_generator_options_bit_exact_vs_reference():
    """SsdAnchorGenerator(min_size / max_size, step, offset, num_branches, flip=False) and build_anchor_generators(sizes, steps,
    num_branches) on the GPU against the reference's own outputs (tests/golden/anchor_options.npz)."""
    import os
    from conftest import GOLDEN
    from test_oracle_golden import ANCHOR_OPTION_CASES
    from single_shot_detection_amd.detection.anchor_generators import ssd
    g = np.load(os.path.join(GOLDEN, 'anchor_options.npz'))
    for name, (kw, img_wh, fmap_wh) in ANCHOR_OPTION_CASES.items():
        gen = ssd.SsdAnchorGenerator(**kw)
        assert gen.num_boxes == int(g[name + '_num_boxes'])
        img = torch.empty((1, 3, img_wh[1], img_wh[0]), device='cuda')
        got = gen.generate(img, (fmap_wh[1], fmap_wh[0])).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), g[name].view(np.uint32)), name
    gens = ssd.build_anchor_generators(num_scales=3, sizes=[30, 60, 111, 162], aspect_ratios=[[1.0, 2.0]] * 3, steps=[8, 16, 32], num_branches=[1, 2, 1])
    img = torch.empty((1, 3, 300, 300), device='cuda')
    got = torch.cat([gn.generate(img, (h, w)).reshape(-1) for gn, (w, h) in zip(gens, [(38, 38), (19, 19), (10, 10)])]).view(-1, 4).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), g['builder'].view(np.uint32))
    with pytest.raises(NotImplementedError):
        ssd.SsdAnchorGenerator([1.0, 2.0], min_scale=0.2)     # no maximum: the reference's own code fails on it (ssd.py:135)


def test_packed_ground_truth_with_padding_equals_the_list_form():
    """target_assigner.PackedGroundTruth (static buffers of fixed capacity for a captured HIP graph): rows past the last image's are padding
    -- NaNs here -- and change nothing; update_ refills the same buffers with another batch (an empty image included)."""
    from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth
    cfg = syn.CONFIGS['ssd_300_vgg16_voc']
    anchors = torch.from_numpy(load_golden('ssd_300_vgg16_voc')['anchors']).cuda()
    ta = TargetAssigner(cfg['matched'], cfg['unmatched'])
    gt_a = [torch.from_numpy(x) for x in syn.make_ground_truth(4, cfg['size'], cfg['num_classes'], seed=4)]
    gt_b = [torch.from_numpy(x) for x in syn.make_ground_truth(4, cfg['size'], cfg['num_classes'], seed=5)]
    gt_b[2] = torch.zeros((0, 6))
    cap = max(sum(len(g) for g in gt_a), sum(len(g) for g in gt_b)) + 5
    packed = PackedGroundTruth.from_list(gt_a, anchors.device, capacity=cap)
    packed.rows[int(packed.offsets[-1]):] = float('nan')
    assert len(packed) == 4
    for gt in (gt_a, gt_b):
        if gt is gt_b:
            packed.update_(gt_b)
            packed.rows[int(packed.offsets[-1]):] = float('nan')
        ref_t, ref_idx = ta.encode_ground_truth(gt, anchors, return_box_idx=True)
        t, idx = ta.encode_ground_truth(packed, anchors, return_box_idx=True)
        assert torch.equal(idx, ref_idx) and np.array_equal(bits(t), bits(ref_t))
    with pytest.raises(ValueError):
        PackedGroundTruth.from_list(gt_a, anchors.device, capacity=1)
