"""CPU: host-side logic of the mirrored API that needs no GPU (constructor quirks, builders, packing, error paths)."""
import functools

import numpy as np
import pytest
import torch

from single_shot_detection_amd import _lib, synthetic as syn
from single_shot_detection_amd.bf.modules import losses
from single_shot_detection_amd.detection import detector_builder, matcher, sampler, target_assigner
from single_shot_detection_amd.detection.box_coder import BoxCoder
from single_shot_detection_amd.detection.losses.multibox_loss import MultiboxLoss, _unwrap_sampler
from single_shot_detection_amd.detection.postprocessor import Postprocessor
from single_shot_detection_amd.utils import filter_kwargs, get_ctor


def test_constants_match_reference():
    assert (matcher.NOT_MATCHED, matcher.IGNORE) == (-2, -1)                       # detection/matcher.py:4-5
    assert (target_assigner.LOC_INDEX_START, target_assigner.LOC_INDEX_END, target_assigner.CLASS_INDEX,
            target_assigner.SCORE_INDEX, target_assigner.TARGET_SIZE) == (0, 4, 4, 5, 6)     # target_assigner.py:7-11
    assert target_assigner.NEGATIVE_CLASS == 0 and target_assigner.IGNORE_CLASS == -1


def test_filter_kwargs_drops_unnamed_keywords_even_with_var_kwargs():
    def f(a, **kwargs):
        return a, kwargs
    assert filter_kwargs(f)(1, b=2) == (1, {})     # bf/utils/misc_utils.py:22-26 behaviour
    assert get_ctor(losses, 'SigmoidFocalLoss')(reduction='sum', gamma=1.5).reduction == 'mean'
    assert get_ctor(losses, 'CrossEntropyLoss')(reduction='sum', ignore_index=-1, name='x').reduction == 'sum'


def test_multibox_loss_constructor_surface():
    bc = BoxCoder(10.0, 5.0)
    smp = functools.partial(sampler.hard_negative_mining, negative_per_positive_ratio=3, min_negative_per_image=5)
    ce = MultiboxLoss(smp, bc, {'name': 'CrossEntropyLoss'}, {'name': 'SmoothL1Loss'})
    assert ce.cls_kind == 0 and ce.smooth_l1_beta == 1.0 and not ce.multiclass
    fo = MultiboxLoss(sampler.naive_sampler, bc, {'name': 'SigmoidFocalLoss', 'gamma': 2.0, 'alpha': 0.25}, {'name': 'SmoothL1Loss'})
    assert fo.cls_kind == 1 and fo.focal_reduce_mean == 1 and fo.multiclass
    fn, kw = _unwrap_sampler(smp)
    assert fn is sampler.hard_negative_mining and kw == {'negative_per_positive_ratio': 3, 'min_negative_per_image': 5}
    sf = MultiboxLoss(smp, bc, {'name': 'SoftmaxFocalLoss', 'gamma': 1.5}, {'name': 'SmoothL1Loss'})
    assert sf.cls_kind == 2 and sf.focal_alpha == -1.0 and sf.focal_reduce_mean == 1     # reduction='sum' dropped here too
    assert MultiboxLoss(smp, bc, {'name': 'CrossEntropyWithSoftTargetsLoss', 'epsilon': 0.1}, {'name': 'SmoothL1Loss'}).cls_kind == 3
    assert MultiboxLoss(smp, bc, {'name': 'BinaryCrossEntropyWithSoftTargetsLoss'}, {'name': 'SmoothL1Loss'}).cls_kind == 4
    gi = MultiboxLoss(smp, bc, {'name': 'CrossEntropyLoss'}, {'name': 'GeneralizedIoULoss'})
    assert gi.loc_kind == 1 and gi.iou_loss
    with pytest.raises(AttributeError):
        MultiboxLoss(smp, bc, {'name': 'NoSuchLoss'}, {'name': 'SmoothL1Loss'})


def test_postprocessor_constructor_surface():
    with pytest.raises(ValueError):
        Postprocessor(BoxCoder(10., 5.), 0.01, {'max_per_class': 100, 'overlap_threshold': .45}, 'TANH')
    Postprocessor(BoxCoder(10., 5.), 0.01, {'overlap_threshold': .45}, 'SOFTMAX')            # max_per_class=None: every candidate enters NMS
    p = Postprocessor(BoxCoder(10., 5.), 0.01, {'overlap_threshold': .45, 'soft': True, 'sigma': 0.3}, 'SOFTMAX')   # ... with soft-NMS too (box_utils.py:166)
    assert p.soft and p.sigma == 0.3


def test_product_path_refuses_cpu_tensors():
    with pytest.raises(_lib.SsdkError):
        target_assigner.TargetAssigner(0.5, 0.5).encode_ground_truth([torch.zeros((1, 6))], torch.zeros((4, 4)))
    with pytest.raises(_lib.SsdkError):
        BoxCoder(10., 5.).decode_box(torch.zeros((1, 4, 4)), torch.ones((4, 4)))


def test_pack_ground_truth_layout():
    gt = [torch.tensor([[1., 2., 3., 4., 5., 1., 0.]]), torch.zeros((0, 7)), torch.tensor([[0., 0., 9., 9., 2., 1., 1.], [1., 1., 2., 2., 3., .5, 0.]])]
    rows, offs, total = target_assigner.pack_ground_truth(gt, torch.device('cpu'))
    assert total == 3 and offs.tolist() == [0, 1, 1, 3] and rows.shape == (3, 6)
    assert rows[2].tolist() == [1., 1., 2., 2., 3., .5]


def test_builder_shapes_and_state_dict_keys():
    heads = detector_builder.get_heads([16, 32], [4, 6], 21, score_head_bias_init=-4.6)
    keys = set(heads.state_dict().keys())
    assert {'0.score.weight', '0.score.bias', '0.loc.weight', '1.loc.bias'} <= keys     # heads.<i>.score|loc.weight|bias
    assert heads[0]['score'].weight.shape == (84, 16, 3, 3) and heads[1]['loc'].weight.shape == (24, 32, 3, 3)
    assert heads[0]['score'].weight.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(heads[0]['score'].bias, torch.full((84,), -4.6))
    extras = detector_builder.get_extras([512], layers=(('s', 512), ('s', 256)))
    assert extras[0][0].conv.weight.shape == (256, 512, 1, 1) and extras[0][1].conv.stride == (2, 2)
    assert extras[1][1].conv.weight.shape == (256, 128, 3, 3)
    with pytest.raises(_lib.SsdkError):      # the tail runs on libssdk: CPU tensors are refused, there is no fallback
        extras[0](torch.zeros((1, 512, 18, 18)))
    assert {'0.0.conv.weight', '0.0.bn.weight', '0.1.conv.weight'} <= set(extras.state_dict().keys())


def test_synthetic_inputs_are_deterministic():
    a = syn.make_ground_truth(4, 300, 81, seed=1)
    b = syn.make_ground_truth(4, 300, 81, seed=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert all(x.shape[1] == 6 and (x[:, 2] <= 299).all() and (x[:, 4] >= 1).all() and (x[:, 4] <= 80).all() for x in a)
    from conftest import CONFIG_NAMES
    assert [syn.num_anchors(syn.CONFIGS[n]) for n in CONFIG_NAMES] == [2268, 8108, 24564, 47961, 24528]


def test_convert_sync_batchnorm_marks_hot_path_norms_and_converts_the_rest():
    """distributed.convert_sync_batchnorm (the role of apex convert_syncbn_model, detection/init.py:85): BatchNorm2d inside the
    hot-path blocks stays a BatchNorm2d (same state_dict keys) marked with the process group, anything else becomes SyncBatchNorm."""
    import torch.nn as nn
    from single_shot_detection_amd import ops
    from single_shot_detection_amd.bf.modules.conv import Conv2dBn
    from single_shot_detection_amd.detection.modules.predictors import SharedConvPredictor
    from single_shot_detection_amd.distributed import convert_sync_batchnorm
    m = nn.Sequential(nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8)), Conv2dBn(8, 16, 1), SharedConvPredictor([16], [3], 4, False, num_layers=1, num_channels=16))
    keys = sorted(m.state_dict())
    convert_sync_batchnorm(m)
    assert sorted(m.state_dict()) == keys
    assert isinstance(m[0][1], nn.SyncBatchNorm)
    assert type(m[1].bn) is nn.BatchNorm2d and ops.sync_group_of(m[1].bn) == (None,)
    for head in ('score', 'loc'):
        assert all(type(n) is nn.BatchNorm2d and ops.sync_group_of(n) == (None,) for n in m[2].norms[head][0])
