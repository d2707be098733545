"""Samplers -- mirror of detection/sampler.py:9-25 on libssdk (csrc/loss.hip).

Signature kept: ``f(predictions [B,A,C], target_classes [B,A]) -> bool [B,A]``.  MultiboxLoss recognises these two
functions (also wrapped in ``functools.partial``) and runs them fused with the loss; any other callable is simply
called and its mask handed to the loss kernel.
"""
import torch

from .. import _lib
from .target_assigner import NEGATIVE_CLASS, IGNORE_CLASS  # noqa: F401


def _class_column(target_classes):
    """float32 class values + element stride for the kernel (a [B,A,6] target's column 4 is used in place)."""
    if target_classes.dtype == torch.float32 and target_classes.dim() == 2 and target_classes.stride(1) in (1, 6) \
            and target_classes.stride(0) == target_classes.size(1) * target_classes.stride(1):
        return target_classes, target_classes.stride(1)
    t = target_classes.to(torch.float32).contiguous()
    return t, 1


def loss_workspace(batch, anchors, classes, device):
    need = _lib.lib().ssdk_multibox_loss_workspace_bytes(batch, anchors, classes)
    return torch.empty((need,), dtype=torch.uint8, device=device)


def naive_sampler(predictions, target_classes):
    _lib.require_cuda(target_classes)
    cls, stride = _class_column(target_classes)
    B, A = target_classes.shape[:2]
    mask = torch.empty((B, A), dtype=torch.uint8, device=target_classes.device)
    _lib.check(_lib.lib().ssdk_naive_sampler(cls.data_ptr(), stride, B, A, _lib.ptr(mask), _lib.current_stream()),
               'ssdk_naive_sampler')
    return mask.bool()


def hard_negative_mining(predictions, target_classes, negative_per_positive_ratio, min_negative_per_image, _workspace=None):
    _lib.require_cuda(predictions, target_classes)
    B, A = target_classes.shape[:2]
    scores = predictions.contiguous().float()
    C = scores.numel() // (B * A)
    cls, stride = _class_column(target_classes)
    ws = _workspace if _workspace is not None else loss_workspace(B, A, C, scores.device)
    mask = torch.empty((B, A), dtype=torch.uint8, device=scores.device)
    _lib.check(_lib.lib().ssdk_hard_negative_mining(_lib.ptr(scores), cls.data_ptr(), stride, B, A, C,
                                                    float(negative_per_positive_ratio), int(min_negative_per_image),
                                                    _lib.ptr(mask), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               'ssdk_hard_negative_mining')
    return mask.bool()
