"""Loss descriptors -- mirror of bf/modules/losses.py for the classes the hot path can select by name.

The arithmetic lives in csrc/loss.hip; these classes only carry the constructor surface (so that
``get_ctor``/``filter_kwargs`` behave exactly like the reference's, including the dropped ``reduction`` keyword
for classes whose ``__init__`` takes ``**kwargs`` -- SURVEY.md §8a L1) and the attributes MultiboxLoss inspects
(``MULTICLASS``, ``SOFT_TARGET``, ``IOU_LOSS``, ``reduction``).
"""
import torch.nn as nn
from torch.nn.modules.loss import CrossEntropyLoss, SmoothL1Loss  # noqa: F401  (losses.py:4 re-exports torch's)


class _Loss(nn.Module):
    def __init__(self, reduction='mean', epsilon=0.0):  # losses.py:8-17
        super(_Loss, self).__init__()
        if reduction not in ['mean', 'sum', 'none']:
            raise ValueError(f'Wrong value for reduction: {reduction}')
        assert 0.0 <= epsilon < 1
        self.reduction = reduction
        self.epsilon = epsilon


class SigmoidFocalLoss(_Loss):
    MULTICLASS = True

    def __init__(self, gamma=2.0, alpha=0.25, **kwargs):  # losses.py:37-40
        super(SigmoidFocalLoss, self).__init__(**kwargs)
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, prediction, target):
        raise RuntimeError('SigmoidFocalLoss is evaluated inside ssdk_multibox_loss_fwd (csrc/loss.hip)')


class SoftmaxFocalLoss(_Loss):
    def __init__(self, gamma=0.0, alpha=None, ignore_index=-100, **kwargs):  # losses.py:57-61
        super(SoftmaxFocalLoss, self).__init__(**kwargs)
        self.gamma = gamma
        self.alpha = alpha
        self.ignore_index = ignore_index


class CrossEntropyWithSoftTargetsLoss(_Loss):  # losses.py:80-93
    SOFT_TARGET = True


class BinaryCrossEntropyWithSoftTargetsLoss(_Loss):  # losses.py:95-106
    SOFT_TARGET = True
    MULTICLASS = True


class GeneralizedIoULoss(_Loss):  # losses.py:109-114
    IOU_LOSS = True
