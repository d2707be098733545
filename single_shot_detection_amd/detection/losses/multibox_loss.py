"""MultiboxLoss -- mirror of detection/losses/multibox_loss.py:10-94 on libssdk (csrc/loss.hip)."""
import ctypes
import functools

import torch
import torch.nn as nn

from ... import _lib
from ...bf.modules import losses
from ...utils import get_ctor
from .. import sampler as _sampler
from ..target_assigner import LOC_INDEX_START, LOC_INDEX_END, CLASS_INDEX, SCORE_INDEX, IGNORE_CLASS, NEGATIVE_CLASS  # noqa: F401

SSDK_CLS_CROSS_ENTROPY = 0
SSDK_CLS_SIGMOID_FOCAL = 1
SSDK_CLS_SOFTMAX_FOCAL = 2
SSDK_CLS_CE_SOFT = 3
SSDK_CLS_BCE_SOFT = 4
SSDK_LOC_SMOOTH_L1 = 0
SSDK_LOC_GIOU = 1


_NO_ROW_HINT = bool(__import__('os').environ.get('SSDK_NO_ROW_HINT'))   # (measurement knob: the heads scan dscores as before round 4)


def _unwrap_sampler(fn):
    """(base function, kwargs) of a possibly functools.partial-wrapped sampler (detection/init.py:90-92)."""
    kwargs = {}
    while isinstance(fn, functools.partial):
        kwargs = {**fn.keywords, **kwargs}
        if fn.args:
            return fn, {}
        fn = fn.func
    return fn, kwargs


class _MultiboxLossFn(torch.autograd.Function):
    """(class_loss, loc_loss) = f(scores, locs); target is mutated in place like the reference does."""

    @staticmethod
    def forward(ctx, scores, locs, anchors, target, module):
        lib = _lib.lib()
        B, A = target.shape[:2]
        C = scores.numel() // (B * A)
        dev = scores.device
        ws = _sampler.loss_workspace(B, A, C, dev)
        cls_col = target[..., CLASS_INDEX]
        fn, kw = _unwrap_sampler(module.sampler)
        lse_valid = 0
        if fn is _sampler.hard_negative_mining and {'negative_per_positive_ratio', 'min_negative_per_image'} <= set(kw):
            mask = torch.empty((B, A), dtype=torch.uint8, device=dev)
            _lib.check(lib.ssdk_hard_negative_mining(_lib.ptr(scores), cls_col.data_ptr(), 6, B, A, C,
                                                     float(kw['negative_per_positive_ratio']),
                                                     int(kw['min_negative_per_image']), _lib.ptr(mask), _lib.ptr(ws),
                                                     ws.numel(), _lib.current_stream()), 'ssdk_hard_negative_mining')
            lse_valid = 1 if module.cls_kind in (SSDK_CLS_CROSS_ENTROPY, SSDK_CLS_SOFTMAX_FOCAL) else 0
        elif fn is _sampler.naive_sampler:
            mask = torch.empty((B, A), dtype=torch.uint8, device=dev)
            _lib.check(lib.ssdk_naive_sampler(cls_col.data_ptr(), 6, B, A, _lib.ptr(mask), _lib.current_stream()),
                       'ssdk_naive_sampler')
        else:  # any user sampler with the reference's signature (multibox_loss.py:58)
            mask = module.sampler(scores.view(B, A, C), cls_col.long()).to(torch.uint8).contiguous()
        out3 = torch.empty((3,), dtype=torch.float32, device=dev)
        params = module.loss_params()
        _lib.check(lib.ssdk_multibox_loss_fwd(ctypes.byref(params), _lib.ptr(scores), _lib.ptr(locs), _lib.ptr(anchors),
                                              _lib.ptr(target), _lib.ptr(mask), B, A, C, lse_valid, _lib.ptr(out3), _lib.ptr(ws),
                                              ws.numel(), _lib.current_stream()), 'ssdk_multibox_loss_fwd')
        ctx.save_for_backward(scores, locs, anchors, target, mask, ws)
        ctx.module = module
        ctx.shape = (B, A, C)
        ctx.sparse_rows = fn is _sampler.hard_negative_mining or fn is _sampler.naive_sampler   # (the gradient rows of the anchors that were not sampled are zeros)
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)   # (an output nobody differentiated arrives as None, not as a zero tensor made by a fill launch)
        return out3[0], out3[1], out3[2], mask   # loss = class_loss + loc_loss (summed by the kernel: multibox_loss.py:93), class_loss, loc_loss

    @staticmethod
    def backward(ctx, g_total, g_class, g_loc, _g_mask):
        scores, locs, anchors, target, mask, ws = ctx.saved_tensors
        module = ctx.module
        B, A, C = ctx.shape
        if g_total is None and g_class is None and g_loc is None:
            return None, None, None, None, None
        # the usual case: only `loss` was differentiated -- its gradient (one device scalar) goes to the kernel as it is; anything else
        # (class_loss / loc_loss used on their own in the same backward pass) is combined into the two-element form first
        single = g_class is None and g_loc is None
        if single:
            grad_out = g_total.float().reshape(1).contiguous()
        else:
            zero = scores.new_zeros(())
            gt = zero if g_total is None else g_total.float().reshape(())
            grad_out = torch.stack([gt + (zero if g_class is None else g_class.float().reshape(())),
                                    gt + (zero if g_loc is None else g_loc.float().reshape(()))]).contiguous()
        dscores = torch.empty_like(scores)
        dlocs = torch.empty_like(locs)
        params = module.loss_params()
        # under hard-negative mining a few % of the anchors carry a gradient: the kernel also writes which (row_mask), and the heads'
        # backward of this pass -- if it receives these very tensors -- reads only those rows instead of scanning all of dscores
        row_mask = torch.empty((B, A), dtype=torch.uint8, device=scores.device) if ctx.sparse_rows and not _NO_ROW_HINT else None
        _lib.check(_lib.lib().ssdk_multibox_loss_bwd_ex(ctypes.byref(params), _lib.ptr(scores), _lib.ptr(locs), _lib.ptr(anchors),
                                                        _lib.ptr(target), _lib.ptr(mask), _lib.ptr(grad_out), 1 if single else 0, B, A, C, _lib.ptr(dscores),
                                                        _lib.ptr(dlocs), _lib.ptr(row_mask), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                   'ssdk_multibox_loss_bwd')
        if row_mask is not None:
            from ..modules import heads as heads_mod
            heads_mod.set_row_hint(dscores, dlocs, row_mask)
        return dscores, dlocs, None, None, None


class MultiboxLoss(nn.Module):
    def __init__(self,
                 sampler,
                 box_coder,
                 classification_loss,
                 localization_loss,
                 classification_weight=1.0,
                 localization_weight=1.0):
        super(MultiboxLoss, self).__init__()
        self.sampler = sampler
        self.box_coder = box_coder

        # multibox_loss.py:23-30 -- same constructor dance, so the same keywords survive filter_kwargs
        ClassificationLoss = get_ctor(losses, classification_loss['name'])
        self.classification_loss = ClassificationLoss(reduction='sum', ignore_index=IGNORE_CLASS, **classification_loss)
        self.soft_target = getattr(self.classification_loss, 'SOFT_TARGET', False)
        self.multiclass = getattr(self.classification_loss, 'MULTICLASS', False)
        LocalizationLoss = get_ctor(losses, localization_loss['name'])
        self.localization_loss = LocalizationLoss(reduction='sum', **localization_loss)
        self.iou_loss = getattr(self.localization_loss, 'IOU_LOSS', False)

        self.classification_weight = classification_weight
        self.localization_weight = localization_weight

        cl = self.classification_loss
        self.focal_gamma, self.focal_alpha, self.focal_reduce_mean, self.soft_epsilon = 2.0, 0.25, 0, 0.0
        if isinstance(cl, losses.CrossEntropyLoss):
            if cl.reduction != 'sum' or cl.ignore_index != IGNORE_CLASS or cl.weight is not None or \
                    getattr(cl, 'label_smoothing', 0.0) != 0.0:
                raise NotImplementedError('CrossEntropyLoss: only reduction=sum, ignore_index=-1, no weights/smoothing')
            self.cls_kind = SSDK_CLS_CROSS_ENTROPY
        elif isinstance(cl, losses.SigmoidFocalLoss):
            if cl.reduction not in ('mean', 'sum'):
                raise NotImplementedError("SigmoidFocalLoss: reduction must be 'mean' or 'sum'")
            self.cls_kind = SSDK_CLS_SIGMOID_FOCAL
            self.focal_gamma, self.focal_alpha = float(cl.gamma), float(cl.alpha)
            self.focal_reduce_mean = 1 if cl.reduction == 'mean' else 0
        elif isinstance(cl, losses.SoftmaxFocalLoss):
            if cl.reduction not in ('mean', 'sum') or cl.ignore_index != IGNORE_CLASS:
                raise NotImplementedError("SoftmaxFocalLoss: reduction must be 'mean' or 'sum' and ignore_index -1")
            self.cls_kind = SSDK_CLS_SOFTMAX_FOCAL
            self.focal_gamma = float(cl.gamma)
            self.focal_alpha = -1.0 if cl.alpha is None else float(cl.alpha)
            self.focal_reduce_mean = 1 if cl.reduction == 'mean' else 0
        elif isinstance(cl, (losses.CrossEntropyWithSoftTargetsLoss, losses.BinaryCrossEntropyWithSoftTargetsLoss)):
            if cl.reduction != 'sum':
                raise NotImplementedError(f'{type(cl).__name__}: only reduction=sum')
            self.cls_kind = SSDK_CLS_BCE_SOFT if self.multiclass else SSDK_CLS_CE_SOFT   # multibox_loss.py:64 tests multiclass first
            self.soft_epsilon = float(cl.epsilon)
        else:
            raise NotImplementedError(f'classification loss {type(cl).__name__} is not on the GPU path')
        ll = self.localization_loss
        self.smooth_l1_beta = 1.0
        if isinstance(ll, losses.SmoothL1Loss) and ll.reduction == 'sum':
            self.loc_kind = SSDK_LOC_SMOOTH_L1
            self.smooth_l1_beta = float(getattr(ll, 'beta', 1.0))
        elif isinstance(ll, losses.GeneralizedIoULoss) and ll.reduction == 'sum':
            self.loc_kind = SSDK_LOC_GIOU
        else:
            raise NotImplementedError(f'localization loss {type(ll).__name__}(reduction={ll.reduction}) is not on the GPU path')
        self.last_sampled_mask = None

    def loss_params(self):
        return _lib.LossParams(self.cls_kind, self.loc_kind, self.focal_gamma, self.focal_alpha, self.focal_reduce_mean,
                               self.soft_epsilon, float(self.classification_weight), float(self.localization_weight),
                               float(self.box_coder.xy_scale), float(self.box_coder.wh_scale), float(self.box_coder.eps),
                               self.smooth_l1_beta)

    def forward(self, pred, anchors, target):
        """
        Args:
            pred: tuple of
                torch.tensor(:shape [Batch, AnchorBoxes * Classes])
                torch.tensor(:shape [Batch, AnchorBoxes * 4])
            target: torch.tensor(:shape [Batch, AnchorBoxes, 6])  -- columns 0..3 are overwritten with the encoded
                regression targets, exactly like multibox_loss.py:81-82
        Returns:
            losses: (loss, class_loss, loc_loss) 0-dim tensors with autograd
        """
        scores, locs = pred
        _lib.require_cuda(scores, locs, anchors, target)
        if target.dtype != torch.float32 or not target.is_contiguous():
            raise ValueError('target must be a contiguous float32 [Batch, AnchorBoxes, 6] tensor')
        scores = scores.float().contiguous()
        locs = locs.float().contiguous()
        anchors = anchors.float().contiguous()
        loss, class_loss, loc_loss, mask = _MultiboxLossFn.apply(scores, locs, anchors, target, self)
        self.last_sampled_mask = mask
        return loss, class_loss, loc_loss
