"""Loss modules -- mirror of bf/modules/losses.py for the classes the hot path can select by name.

Inside MultiboxLoss the arithmetic is fused into ssdk_multibox_loss_fwd / _bwd (csrc/loss.hip) and these classes only carry the
constructor surface (so that ``get_ctor``/``filter_kwargs`` behave exactly like the reference's, including the dropped ``reduction``
keyword for classes whose ``__init__`` takes ``**kwargs`` -- SURVEY.md §8a L1) and the attributes MultiboxLoss inspects
(``MULTICLASS``, ``SOFT_TARGET``, ``IOU_LOSS``, ``reduction``).

On their own (``forward(prediction, target)``, losses.py:42-114) they run the SAME kernels on the [rows, classes] slice they are given:
the rows become the anchors of a one-image batch in which every row is sampled, the target rows -- which MultiboxLoss builds with at most
one non-zero entry, multibox_loss.py:62-74 -- go back to the (class, score) columns the kernel reads, and ``reduce_mean = 2`` asks for
the plain sums instead of MultiboxLoss's division by the positives.  Limits, each refused loudly: CUDA tensors, ``reduction`` 'sum' or
'mean' ('none' has no per-row output in the fused kernel), dense targets with at most one non-zero entry per row, SoftmaxFocalLoss's
``ignore_index`` mapped onto the kernel's -1.  GeneralizedIoULoss (losses.py:109-114) is 1 - box_utils.generalized_iou on
csrc/boxes.hip, forward only (no autograd: MultiboxLoss's GIoU term has its backward inside the fused kernel).
"""
import ctypes

import torch
import torch.nn as nn
from torch.nn.modules.loss import CrossEntropyLoss, SmoothL1Loss  # noqa: F401  (losses.py:4 re-exports torch's)

SSDK_CLS_SIGMOID_FOCAL = 1
SSDK_CLS_SOFTMAX_FOCAL = 2
SSDK_CLS_CE_SOFT = 3
SSDK_CLS_BCE_SOFT = 4


class _SliceLossFn(torch.autograd.Function):
    """class-loss sum of ssdk_multibox_loss_fwd over a [rows, classes] slice (every row sampled, no box term)."""

    @staticmethod
    def forward(ctx, prediction, cls_col, score_col, kind, gamma, alpha, epsilon):
        from ... import _lib
        from ...detection import sampler as _sampler
        lib = _lib.lib()
        P, C = prediction.shape
        dev = prediction.device
        scores = prediction.float().contiguous()
        target = torch.zeros((1, P, 6), dtype=torch.float32, device=dev)
        target[0, :, 4] = cls_col
        target[0, :, 5] = score_col
        locs = torch.zeros((1, P * 4), dtype=torch.float32, device=dev)
        anchors = torch.ones((P, 4), dtype=torch.float32, device=dev)   # (box term: weight 0, and finite: log(0 / 1 + eps) with eps = 1)
        sampled = torch.ones((1, P), dtype=torch.uint8, device=dev)
        ws = _sampler.loss_workspace(1, P, C, dev)
        params = _lib.LossParams(kind, 0, float(gamma), float(alpha), 2, float(epsilon), 1.0, 0.0, 10.0, 5.0, 1.0, 1.0)
        out3 = torch.empty((3,), dtype=torch.float32, device=dev)
        _lib.check(lib.ssdk_multibox_loss_fwd(ctypes.byref(params), _lib.ptr(scores), _lib.ptr(locs), _lib.ptr(anchors), _lib.ptr(target),
                                              _lib.ptr(sampled), 1, P, C, 0, _lib.ptr(out3), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                   'ssdk_multibox_loss_fwd')
        ctx.save_for_backward(scores, locs, anchors, target, sampled, ws)
        ctx.params = params
        ctx.dtype = prediction.dtype
        return out3[1]

    @staticmethod
    def backward(ctx, grad):
        from ... import _lib
        scores, locs, anchors, target, sampled, ws = ctx.saved_tensors
        P, C = scores.shape
        grad_out = grad.float().reshape(1).expand(2).contiguous()
        dscores = torch.empty_like(scores)
        dlocs = torch.empty_like(locs)
        _lib.check(_lib.lib().ssdk_multibox_loss_bwd_ex(ctypes.byref(ctx.params), _lib.ptr(scores), _lib.ptr(locs), _lib.ptr(anchors),
                                                        _lib.ptr(target), _lib.ptr(sampled), _lib.ptr(grad_out), 0, 1, P, C, _lib.ptr(dscores),
                                                        _lib.ptr(dlocs), None, _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                   'ssdk_multibox_loss_bwd')
        return dscores.to(ctx.dtype), None, None, None, None, None, None


class _Loss(nn.Module):
    def __init__(self, reduction='mean', epsilon=0.0):  # losses.py:8-17
        super(_Loss, self).__init__()
        if reduction not in ['mean', 'sum', 'none']:
            raise ValueError(f'Wrong value for reduction: {reduction}')
        assert 0.0 <= epsilon < 1
        self.reduction = reduction
        self.epsilon = epsilon

    def _check(self, prediction):
        if not prediction.is_cuda:
            raise RuntimeError(f'{type(self).__name__}.forward runs on libssdk: CUDA tensors only')
        if self.reduction == 'none':
            raise NotImplementedError(f"{type(self).__name__}: reduction='none' (the fused kernel has no per-row output)")
        if prediction.dim() != 2:
            raise ValueError(f'{type(self).__name__}: prediction must be [rows, classes]')

    @staticmethod
    def _one_entry_rows(target, what):
        """(column of the non-zero entry or -1, its value) of every row of a dense [rows, classes] target; rows with more than one
        non-zero entry are not what multibox_loss.py:62-74 builds and not what the kernel reads."""
        nz = target != 0
        if bool((nz.sum(-1) > 1).any()):
            raise NotImplementedError(f'{what}: a target row with more than one non-zero entry')
        col = torch.where(nz.any(-1), nz.float().argmax(-1), torch.full((target.shape[0],), -1, device=target.device, dtype=torch.long))
        val = target.float().sum(-1)
        return col, val


class SigmoidFocalLoss(_Loss):
    MULTICLASS = True

    def __init__(self, gamma=2.0, alpha=0.25, **kwargs):  # losses.py:37-40
        super(SigmoidFocalLoss, self).__init__(**kwargs)
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, prediction, target):  # losses.py:42-54; target [rows, classes], column j = class j + 1 (multibox_loss.py:64-67)
        self._check(prediction)
        col, val = self._one_entry_rows(target, 'SigmoidFocalLoss')
        total = _SliceLossFn.apply(prediction, (col + 1).float(), val, SSDK_CLS_SIGMOID_FOCAL, self.gamma, self.alpha, 0.0)
        return total / prediction.shape[0] if self.reduction == 'mean' else total   # (:52 sums a row's classes, _reduce means over the rows)


class SoftmaxFocalLoss(_Loss):
    def __init__(self, gamma=0.0, alpha=None, ignore_index=-100, **kwargs):  # losses.py:57-61
        super(SoftmaxFocalLoss, self).__init__(**kwargs)
        self.gamma = gamma
        self.alpha = alpha
        self.ignore_index = ignore_index

    def forward(self, input_, target):  # losses.py:63-78; target [rows] class ids
        self._check(input_)
        cls = torch.where(target == self.ignore_index, torch.full_like(target, -1), target).float()
        total = _SliceLossFn.apply(input_, cls, torch.zeros_like(cls), SSDK_CLS_SOFTMAX_FOCAL, self.gamma,
                                   -1.0 if self.alpha is None else self.alpha, 0.0)
        return total / target.numel() if self.reduction == 'mean' else total   # (:65 the ignored rows stay in the mean, as zeros)


class CrossEntropyWithSoftTargetsLoss(_Loss):  # losses.py:80-93
    SOFT_TARGET = True

    def forward(self, logits, target):  # target [rows, classes], column j = class j (multibox_loss.py:68-71); an all-zero row = ignored
        self._check(logits)
        col, val = self._one_entry_rows(target, 'CrossEntropyWithSoftTargetsLoss')
        total = _SliceLossFn.apply(logits, col.float(), val, SSDK_CLS_CE_SOFT, 0.0, 0.0, self.epsilon)
        return total / logits.shape[0] if self.reduction == 'mean' else total


class BinaryCrossEntropyWithSoftTargetsLoss(_Loss):  # losses.py:95-106
    SOFT_TARGET = True
    MULTICLASS = True

    def forward(self, logits, target):  # target [rows, classes], column j = class j + 1 (multibox_loss.py:64-67)
        self._check(logits)
        col, val = self._one_entry_rows(target, 'BinaryCrossEntropyWithSoftTargetsLoss')
        total = _SliceLossFn.apply(logits, (col + 1).float(), val, SSDK_CLS_BCE_SOFT, 0.0, 0.0, self.epsilon)
        return total / logits.numel() if self.reduction == 'mean' else total   # (:106 F.binary_cross_entropy_with_logits means over the elements)


class GeneralizedIoULoss(_Loss):  # losses.py:109-114
    IOU_LOSS = True

    def forward(self, boxes, target):
        from ..utils import box_utils
        if self.reduction == 'none':
            raise NotImplementedError("GeneralizedIoULoss: reduction='none'")
        if boxes.requires_grad:
            raise NotImplementedError('GeneralizedIoULoss.forward on its own has no backward; inside MultiboxLoss it is fused (csrc/loss.hip)')
        loss = 1.0 - box_utils.generalized_iou(boxes, target, cartesian=False)
        return loss.mean() if self.reduction == 'mean' else loss.sum()
