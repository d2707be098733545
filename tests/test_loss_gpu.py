"""GPU parity: samplers (S1), MultiboxLoss forward/backward (L1/L2/L3), BoxCoder -- through the C ABI, against the
reference's golden vectors and the oracle.  Tolerances: losses 1e-4 absolute (north_star); masks exact up to
exact-threshold near-ties; gradients rtol 1e-4."""
import functools

import numpy as np
import pytest
import torch

import oracle
from single_shot_detection_amd import synthetic as syn
from single_shot_detection_amd.detection import sampler
from single_shot_detection_amd.detection.box_coder import BoxCoder
from single_shot_detection_amd.detection.losses.multibox_loss import MultiboxLoss
from conftest import CONFIG_NAMES, GOLDEN_BATCH, load_golden, dense_from_rows

pytestmark = pytest.mark.gpu


def make_criterion(kind, ratio=3, min_neg=5):
    box_coder = BoxCoder(xy_scale=10.0, wh_scale=5.0)
    if kind == 'ce_hnm':
        smp = functools.partial(sampler.hard_negative_mining, negative_per_positive_ratio=ratio, min_negative_per_image=min_neg)
        cl = {'name': 'CrossEntropyLoss'}
    else:
        smp = sampler.naive_sampler
        cl = {'name': 'SigmoidFocalLoss', 'gamma': 2.0, 'alpha': 0.25}
    return MultiboxLoss(sampler=smp, box_coder=box_coder, classification_loss=cl,
                        localization_loss={'name': 'SmoothL1Loss'}, classification_weight=1.0, localization_weight=1.0)


def check_mask(mask, ref_mask, bg=None):
    if np.array_equal(mask, ref_mask):
        return
    diff = mask != ref_mask
    assert mask.sum() == ref_mask.sum()
    assert diff.sum() <= 8, diff.sum()
    if bg is not None:
        for i in range(mask.shape[0]):
            d = np.where(diff[i])[0]
            if len(d):
                assert np.ptp(bg[i, d]) <= 2e-5


def test_ctor_quirk_is_reproduced():
    assert make_criterion('focal').classification_loss.reduction == 'mean'
    assert make_criterion('focal').focal_reduce_mean == 1
    assert make_criterion('ce_hnm').classification_loss.reduction == 'sum'


def test_kat_hnm_on_gpu(kats):
    for tag in ('kat8', 'kat9'):
        pred = torch.from_numpy(kats[f'{tag}_pred']).cuda()
        cls = torch.from_numpy(kats[f'{tag}_cls']).cuda()
        m = sampler.hard_negative_mining(pred, cls, 3, 5)
        assert np.array_equal(m.cpu().numpy(), kats[f'{tag}_mask']), tag
        m2 = sampler.naive_sampler(pred, cls)
        assert np.array_equal(m2.cpu().numpy(), (kats[f'{tag}_cls'] != 0) & (kats[f'{tag}_cls'] != -1))


def test_kat_box_coder_on_gpu(kats):
    bc = BoxCoder(10.0, 5.0)
    pri = torch.from_numpy(kats['kat10_priors']).cuda()
    cen = torch.from_numpy(kats['kat10_centroids_inplace']).cuda()
    np.testing.assert_allclose(bc.encode_box(cen, pri).cpu().numpy(), kats['kat10_encode'], rtol=0, atol=3e-6)
    enc = cen.clone()
    r = bc.encode_box(enc, pri, inplace=True)
    assert r.data_ptr() == enc.data_ptr()
    np.testing.assert_allclose(enc.cpu().numpy(), kats['kat10_encode_inplace'], rtol=0, atol=3e-6)
    dec = bc.decode_box(torch.from_numpy(kats['kat10_encode_inplace']).cuda(), pri)
    np.testing.assert_allclose(dec.cpu().numpy(), kats['kat10_decode'], rtol=3e-6, atol=1e-5)


@pytest.mark.parametrize('variant', ['rand', 'trained'])
@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_loss_matches_reference_golden(name, variant):
    cfg = syn.CONFIGS[name]
    g = load_golden(name)
    B, A, Cn = GOLDEN_BATCH[name], g['anchors'].shape[0], cfg['num_classes']
    softmax = cfg['score_converter'] == 'SOFTMAX'
    trained = variant == 'trained'
    logits = syn.make_logits(B, A, Cn, seed=2, trained_like=trained and softmax)
    if trained and not softmax:
        logits = logits - np.float32(4.6)
    locs = syn.make_locs(B, A, seed=3, scale=0.5)
    p = f'loss_{variant}_'
    ref_mask = np.unpackbits(g[p + 'sampled_bits'], axis=1)[:, :A].astype(bool)

    scores_t = torch.from_numpy(logits).cuda().requires_grad_(True)
    locs_t = torch.from_numpy(locs).cuda().requires_grad_(True)
    target = torch.from_numpy(g['match_target'].copy()).cuda()
    anchors = torch.from_numpy(g['anchors']).cuda()
    crit = make_criterion(cfg['loss'])
    loss, class_loss, loc_loss = crit((scores_t, locs_t), anchors, target)
    loss.backward()
    torch.cuda.synchronize()

    _, bg = oracle.hard_negative_mining(logits, g['match_target'], 3, 5, return_bgloss=True)
    check_mask(crit.last_sampled_mask.cpu().numpy().astype(bool), ref_mask, bg)
    vals = np.array([loss.item(), class_loss.item(), loc_loss.item()])
    np.testing.assert_allclose(vals, g[p + 'values'], rtol=0, atol=1e-4)          # north_star: loss within 1e-4
    ref_ds = dense_from_rows(g[p + 'dscores_rows'], g[p + 'dscores_vals'], (B, A, Cn))
    ref_dl = dense_from_rows(g[p + 'dlocs_rows'], g[p + 'dlocs_vals'], (B, A, 4))
    ds = scores_t.grad.view(B, A, Cn).cpu().numpy()
    dl = locs_t.grad.view(B, A, 4).cpu().numpy()
    same = (crit.last_sampled_mask.cpu().numpy().astype(bool) == ref_mask)
    np.testing.assert_allclose(ds[same], ref_ds[same], rtol=1e-4, atol=2e-7)
    np.testing.assert_allclose(dl, ref_dl, rtol=1e-5, atol=1e-8)
    if variant == 'rand':   # the in-place mutation of target[..., 0:4] (multibox_loss.py:81-82)
        enc = target[..., :4].cpu().numpy()
        np.testing.assert_allclose(enc[0, :2048], g['loss_encoded_target_img0_first2k'], rtol=1e-6, atol=2e-5)
        cls = g['match_target'][..., 4]
        np.testing.assert_allclose(enc[cls > 0], g['loss_encoded_target_pos'], rtol=1e-6, atol=2e-5)
        assert np.array_equal(target[..., 4:].cpu().numpy(), g['match_target'][..., 4:])


@pytest.mark.parametrize('name,batch', [('ssd_300_vgg16_voc', 32), ('ssd_512_vgg16_coco', 16), ('retina_rn50_500_coco', 8),
                                        ('retina_rn50_500_coco', 32), ('m2det_512_vgg16_coco', 16)])
def test_loss_full_size_vs_oracle(name, batch):
    cfg = syn.CONFIGS[name]
    g = load_golden(name)
    A, Cn = g['anchors'].shape[0], cfg['num_classes']
    softmax = cfg['score_converter'] == 'SOFTMAX'
    gt = syn.make_ground_truth(batch, cfg['size'], Cn, seed=21, background=softmax)
    target_np = oracle.encode_ground_truth(gt, g['anchors'], cfg['matched'], cfg['unmatched'])
    logits = syn.make_logits(batch, A, Cn, seed=22, trained_like=softmax)
    locs = syn.make_locs(batch, A, seed=23, scale=0.5)
    kind = 'ce' if cfg['loss'] == 'ce_hnm' else 'focal'
    if kind == 'ce':
        ref_mask, bg = oracle.hard_negative_mining(logits, target_np, 3, 5, return_bgloss=True)
    else:
        ref_mask, bg = oracle.naive_sampler(logits, target_np), None

    scores_t = torch.from_numpy(logits).cuda().requires_grad_(True)
    locs_t = torch.from_numpy(locs).cuda().requires_grad_(True)
    target = torch.from_numpy(target_np.copy()).cuda()
    crit = make_criterion(cfg['loss'])
    loss, class_loss, loc_loss = crit((scores_t, locs_t), torch.from_numpy(g['anchors']).cuda(), target)
    (2.0 * class_loss + 0.5 * loc_loss).backward()       # non-unit upstream gradients
    mask = crit.last_sampled_mask.cpu().numpy().astype(bool)
    check_mask(mask, ref_mask, bg)
    tgt_ref = target_np.copy()
    vals, ds, dl = oracle.multibox_loss(logits, locs, g['anchors'], tgt_ref, mask, kind=kind, reduce_mean=True)
    got = np.array([loss.item(), class_loss.item(), loc_loss.item()])
    np.testing.assert_allclose(got, vals, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(scores_t.grad.view(batch, A, Cn).cpu().numpy(), 2.0 * ds, rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(locs_t.grad.view(batch, A, 4).cpu().numpy(), 0.5 * dl, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(target.cpu().numpy(), tgt_ref, rtol=1e-6, atol=2e-5)


def test_hnm_edge_cases():
    """no positives (min_negative only), every negative selected, fractional ratio, heavy ties, an ignore band."""
    rng = np.random.default_rng(5)
    B, A, C = 4, 1000, 7
    logits = rng.standard_normal((B, A, C), dtype=np.float32)
    logits[2] = np.round(logits[2] * 2) / 2          # many exactly equal losses -> index-ordered tie break
    logits[3, :, :] = 0.25                           # every loss identical
    cls = np.zeros((B, A), np.float32)
    cls[1, :400] = rng.integers(1, C, 400)            # 3 * 400 > #neg -> all negatives
    cls[2, ::50] = 3
    cls[2, 1::50] = -1
    cls[3, :10] = 2
    target = np.zeros((B, A, 6), np.float32)
    target[..., 4] = cls
    for ratio, min_neg in ((3, 5), (2.5, 0), (0.5, 7), (3, 2000)):
        ref = oracle.hard_negative_mining(logits, target, ratio, min_neg)
        got = sampler.hard_negative_mining(torch.from_numpy(logits).cuda(), torch.from_numpy(cls).cuda().long(), ratio, min_neg)
        got = got.cpu().numpy()
        # images 0,1: distinct floats -> exact; 2,3: exact as well because both break ties by lower index
        assert np.array_equal(got, ref), (ratio, min_neg, (got != ref).sum(axis=1))

@pytest.mark.parametrize('A', [8732, 20000, 140000])
def test_hnm_ties_across_chunks(A):
    """Heavy ties at anchor counts that span several 1024-anchor chunks of hnm_select_kernel's tie table (9, 20) and one past the
    table (137 chunks: the sequential tie pass): the index-ordered tie break must hold across chunk and wave boundaries."""
    rng = np.random.default_rng(17)
    B, C = 2, 5
    # 24 prototype rows with well separated background losses, dealt at random: equal rows give bit-equal losses in both
    # implementations, different prototypes differ by far more than the rounding of either log-sum-exp
    proto = rng.standard_normal((24, C), dtype=np.float32)
    proto[:, 0] = np.linspace(-3.0, 3.0, 24, dtype=np.float32)
    logits = proto[rng.integers(0, 24, (B, A))]
    logits[1] = 0.5                                                               # every loss identical
    cls = np.zeros((B, A), np.float32)
    cls[0, ::97] = 2
    cls[0, 5::211] = -1
    cls[1, 3::1000] = 1
    target = np.zeros((B, A, 6), np.float32)
    target[..., 4] = cls
    for ratio, min_neg in ((3, 5), (7.5, 0)):
        ref = oracle.hard_negative_mining(logits, target, ratio, min_neg)
        got = sampler.hard_negative_mining(torch.from_numpy(logits).cuda(), torch.from_numpy(cls).cuda().long(), ratio, min_neg).cpu().numpy()
        assert np.array_equal(got, ref), (A, ratio, min_neg, (got != ref).sum(axis=1))


def test_custom_sampler_callable_and_ignore_rows():
    """A user sampler (reference signature) that also samples ignored anchors: CE must skip class -1 rows."""
    rng = np.random.default_rng(8)
    B, A, C = 2, 300, 5
    logits = rng.standard_normal((B, A * C), dtype=np.float32)
    locs = rng.standard_normal((B, A * 4), dtype=np.float32)
    anchors = np.concatenate([rng.uniform(10, 90, (A, 2)), rng.uniform(5, 40, (A, 2))], axis=1).astype(np.float32)
    target = np.zeros((B, A, 6), np.float32)
    target[..., 5] = 1
    pos = rng.random((B, A)) < 0.1
    ign = (~pos) & (rng.random((B, A)) < 0.05)
    box = rng.uniform(0, 60, (B, A, 2)).astype(np.float32)
    target[..., 0:2] = box
    target[..., 2:4] = box + rng.uniform(4, 30, (B, A, 2)).astype(np.float32)
    target[pos, 4] = rng.integers(1, C, pos.sum())
    target[ign, 4] = -1
    target[~pos, 0:4] = 0

    def every_third(predictions, target_classes):
        m = torch.zeros_like(target_classes, dtype=torch.bool)
        m[:, ::3] = True
        return m | (target_classes > 0)

    crit = MultiboxLoss(sampler=every_third, box_coder=BoxCoder(10.0, 5.0), classification_loss={'name': 'CrossEntropyLoss'},
                        localization_loss={'name': 'SmoothL1Loss'}, classification_weight=0.7, localization_weight=1.3)
    s = torch.from_numpy(logits).cuda().requires_grad_(True)
    l = torch.from_numpy(locs).cuda().requires_grad_(True)
    t = torch.from_numpy(target.copy()).cuda()
    loss, cl, ll = crit((s, l), torch.from_numpy(anchors).cuda(), t)
    loss.backward()
    mask = np.zeros((B, A), bool)
    mask[:, ::3] = True
    mask |= target[..., 4] > 0
    tr = target.copy()
    vals, ds, dl = oracle.multibox_loss(logits, locs, anchors, tr, mask, kind='ce', cls_w=0.7, loc_w=1.3)
    np.testing.assert_allclose([loss.item(), cl.item(), ll.item()], vals, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s.grad.view(B, A, C).cpu().numpy(), ds, rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(l.grad.view(B, A, 4).cpu().numpy(), dl, rtol=1e-5, atol=1e-8)


def test_no_positives_divider_is_one():
    B, A, C = 2, 128, 4
    rng = np.random.default_rng(3)
    logits = rng.standard_normal((B, A * C), dtype=np.float32)
    locs = rng.standard_normal((B, A * 4), dtype=np.float32)
    anchors = np.tile(np.array([[50, 50, 20, 20]], np.float32), (A, 1))
    target = np.zeros((B, A, 6), np.float32)
    target[..., 5] = 1
    crit = make_criterion('ce_hnm')
    t = torch.from_numpy(target.copy()).cuda()
    loss, cl, ll = crit((torch.from_numpy(logits).cuda(), torch.from_numpy(locs).cuda()), torch.from_numpy(anchors).cuda(), t)
    mask = oracle.hard_negative_mining(logits, target, 3, 5)
    assert mask.sum() == 5 * B
    vals, _, _ = oracle.multibox_loss(logits, locs, anchors, target.copy(), mask, kind='ce', grads=False)
    np.testing.assert_allclose([loss.item(), cl.item(), ll.item()], vals, rtol=1e-5, atol=1e-5)
    assert ll.item() == 0.0


# ---- the other selectable losses on the GPU, against the reference's goldens -------------------------------
EXTRA = {   # tag: (sampler, classification_loss, localization_loss, classes)
    'softmax_focal': ('hnm', {'name': 'SoftmaxFocalLoss', 'gamma': 2.0, 'alpha': 0.25}, {'name': 'SmoothL1Loss'}, 21),
    'softmax_focal_noalpha': ('hnm', {'name': 'SoftmaxFocalLoss', 'gamma': 1.5}, {'name': 'SmoothL1Loss'}, 21),
    'ce_soft': ('hnm', {'name': 'CrossEntropyWithSoftTargetsLoss'}, {'name': 'SmoothL1Loss'}, 21),
    'ce_soft_eps': ('hnm', {'name': 'CrossEntropyWithSoftTargetsLoss', 'epsilon': 0.1}, {'name': 'SmoothL1Loss'}, 21),
    'bce_soft': ('naive', {'name': 'BinaryCrossEntropyWithSoftTargetsLoss'}, {'name': 'SmoothL1Loss'}, 20),
    'giou': ('hnm', {'name': 'CrossEntropyLoss'}, {'name': 'GeneralizedIoULoss'}, 21),
}


@pytest.mark.parametrize('tag', sorted(EXTRA))
def test_extra_losses_vs_reference_golden(tag):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'losses_extra.npz'))
    smp_name, cl, ll, nc = EXTRA[tag]
    anchors_np = load_golden('ssd_mb2_voc')['anchors']
    B, A = 2, anchors_np.shape[0]
    logits = syn.make_logits(B, A, nc, seed=2)
    locs = syn.make_locs(B, A, seed=3, scale=0.5)
    smp = functools.partial(sampler.hard_negative_mining, negative_per_positive_ratio=3, min_negative_per_image=5) if smp_name == 'hnm' \
        else sampler.naive_sampler
    crit = MultiboxLoss(sampler=smp, box_coder=BoxCoder(10.0, 5.0), classification_loss=cl, localization_loss=ll)
    s_t = torch.from_numpy(logits).cuda().requires_grad_(True)
    l_t = torch.from_numpy(locs).cuda().requires_grad_(True)
    target = torch.from_numpy(g['target'].copy()).cuda()
    loss, class_loss, loc_loss = crit((s_t, l_t), torch.from_numpy(anchors_np).cuda(), target)
    loss.backward()
    ref_mask = np.unpackbits(g[tag + '_sampled_bits'], axis=1)[:, :A].astype(bool)
    mask = crit.last_sampled_mask.cpu().numpy().astype(bool)
    check_mask(mask, ref_mask)
    np.testing.assert_allclose([loss.item(), class_loss.item(), loc_loss.item()], g[tag + '_values'], rtol=3e-6, atol=1e-4)
    same = mask == ref_mask
    ds = s_t.grad.view(B, A, nc).cpu().numpy()
    np.testing.assert_allclose(ds[same], dense_from_rows(g[tag + '_dscores_rows'], g[tag + '_dscores_vals'], (B, A, nc))[same], rtol=3e-4, atol=3e-7)
    np.testing.assert_allclose(l_t.grad.view(B, A, 4).cpu().numpy(), dense_from_rows(g[tag + '_dlocs_rows'], g[tag + '_dlocs_vals'], (B, A, 4)),
                               rtol=3e-4, atol=3e-7)
    assert bool(g[tag + '_target_mutated']) == (not np.array_equal(target.cpu().numpy(), g['target']))


@pytest.mark.parametrize('tag', sorted(EXTRA))
def test_loss_modules_on_their_own_vs_reference_golden(tag):
    """bf/modules/losses.py:34-114 outside MultiboxLoss: forward(prediction, target) on the rows the reference's MultiboxLoss hands its
    classification / localisation module (multibox_loss.py:59-86, rebuilt here from the golden sampled mask and target), against the
    reference's class_loss / loc_loss of tests/golden/losses_extra.npz (x the divider of :88), and the gradient against its dscores."""
    import os
    from conftest import GOLDEN
    from single_shot_detection_amd.bf.modules import losses
    from single_shot_detection_amd.bf.utils import box_utils
    g = np.load(os.path.join(GOLDEN, 'losses_extra.npz'))
    smp_name, cl, ll, nc = EXTRA[tag]
    anchors_np = load_golden('ssd_mb2_voc')['anchors']
    B, A = 2, anchors_np.shape[0]
    logits = torch.from_numpy(syn.make_logits(B, A, nc, seed=2)).cuda().view(B, A, nc)
    locs = torch.from_numpy(syn.make_locs(B, A, seed=3, scale=0.5)).cuda().view(B, A, 4)
    target = torch.from_numpy(g['target'].copy()).cuda()
    mask = torch.from_numpy(np.unpackbits(g[tag + '_sampled_bits'], axis=1)[:, :A].astype(bool)).cuda()
    cls = target[..., 4].long()
    positive = (cls != 0) & (cls != -1)
    divider = float(positive.sum().clamp(min=1))
    values = g[tag + '_values']
    if ll['name'] == 'GeneralizedIoULoss':
        coder = BoxCoder(10.0, 5.0)
        anchors = torch.from_numpy(anchors_np).cuda()
        boxes = box_utils.to_corners(coder.decode_box(locs, anchors))
        module = losses.GeneralizedIoULoss(reduction=str(g[tag + '_loc_reduction']))
        got = module(boxes[positive].view(-1, 4), target[..., :4][positive].view(-1, 4))
        np.testing.assert_allclose(got.item(), values[2] * divider, rtol=1e-5)
        with pytest.raises(NotImplementedError):
            module(boxes[positive].view(-1, 4).clone().requires_grad_(True), target[..., :4][positive].view(-1, 4))
        return
    kw = {k: v for k, v in cl.items() if k != 'name'}
    Loss = getattr(losses, cl['name'])
    reduction = str(g[tag + '_cls_reduction'])
    module = Loss(reduction=reduction, ignore_index=-1, **kw) if cl['name'] == 'SoftmaxFocalLoss' else Loss(reduction=reduction, **kw)
    scores = logits[mask].clone().requires_grad_(True)
    t_cls, t_score = cls[mask], target[..., 5][mask]
    if module.__class__.__dict__.get('MULTICLASS', False):           # multibox_loss.py:64-67
        class_target = torch.zeros_like(scores)
        m = (t_cls != 0) & (t_cls != -1)
        class_target[m, t_cls[m] - 1] = t_score[m]
    elif getattr(module, 'SOFT_TARGET', False):                      # :68-71
        class_target = torch.zeros_like(scores)
        m = t_cls != -1
        class_target[m, t_cls[m]] = t_score[m]
    else:                                                            # :73
        class_target = t_cls.view(-1)
    got = module(scores, class_target.detach())
    np.testing.assert_allclose(got.item(), values[1] * divider, rtol=1e-5)
    got.backward()
    dense = torch.zeros((B, A, nc), device='cuda')
    dense[mask] = scores.grad / divider
    ref = dense_from_rows(g[tag + '_dscores_rows'], g[tag + '_dscores_vals'], (B, A, nc))
    np.testing.assert_allclose(dense.cpu().numpy(), ref, rtol=3e-4, atol=3e-7)
    with pytest.raises(NotImplementedError):
        type(module)(reduction='none')(scores.detach(), class_target.detach())
    if class_target.dim() == 2:
        bad = class_target.detach().clone()
        bad[0, :2] = 0.5
        with pytest.raises(NotImplementedError):
            module(scores.detach(), bad)
