"""mean_average_precision -- mirror of detection/metrics/mean_average_precision.py:10-116 on libssdk (csrc/metrics.hip).

Same signature and return value (a python float); predictions and ground truth may live on the host (the reference's
bf/eval.py:63-64 moves them there) or already on the GPU -- they are packed once and evaluated on the device."""
import logging

import numpy as np
import torch

from ... import _lib

DIFFICULT_INDEX = 6   # bf/datasets/detection_dataset.py:15


def average_precisions(predictions, gts, num_classes, iou_threshold, voc=False, device=None):
    """-> (mAP: float, ap: torch.tensor [num_classes] on the host, NaN for classes without counted ground truth)"""
    lib = _lib.lib()
    if device is None:
        device = predictions.device if predictions.is_cuda else torch.device('cuda', torch.cuda.current_device())
    pred = predictions.to(device=device, dtype=torch.float32).contiguous()
    assert pred.dim() == 2 and pred.size(1) == 7, 'predictions: [NumBoxes, 7] = image id, box, class, score'
    stride = int(gts[0].size(1)) if len(gts) else 6
    counts = [int(g.size(0)) for g in gts]
    total = sum(counts)
    rows = torch.cat([g.reshape(-1, stride).to(torch.float32) for g in gts], dim=0).to(device).contiguous() if total else \
        torch.zeros((0, stride), dtype=torch.float32, device=device)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(device)
    n = pred.size(0)
    ws = torch.empty((max(lib.ssdk_mean_average_precision_workspace_bytes(n, total, num_classes), 256),), dtype=torch.uint8, device=device)
    ap = torch.empty((num_classes,), dtype=torch.float32, device=device)
    mean = torch.empty((1,), dtype=torch.float64, device=device)
    _lib.check(lib.ssdk_mean_average_precision(_lib.ptr(pred), n, _lib.ptr(rows), stride, _lib.ptr(offs), len(gts), total, num_classes,
                                               float(iou_threshold), int(bool(voc)), _lib.ptr(ap), _lib.ptr(mean), _lib.ptr(ws), ws.numel(),
                                               _lib.current_stream()), 'ssdk_mean_average_precision')
    return float(mean.item()), ap.cpu()


def mean_average_precision(predictions, gts, class_labels, iou_threshold, voc=False, verbose=True):
    """
    Args:
        predictions: torch.tensor(:shape [NumBoxes, 7] ~ {[0] - image_id, [1-4] - box, [5] - class, [6] - score})
        gts: list(:len NumImages) ~ torch.tensor(:shape [NumBoxes_i, NumAttributes])
        class_labels: dict(:keys ClassId, :values ClassName)
        iou_threshold: float
        voc: bool
        verbose: bool
    Returns:
        mAP: float
    """
    num_classes = max([int(k) for k in class_labels] + [0]) + 1
    for g in gts:   # class ids outside class_labels would be silently dropped on the device: keep the reference's KeyError visible
        if g.numel() and int(g[:, 4].max().item()) >= num_classes:
            num_classes = int(g[:, 4].max().item()) + 1
    mAP, ap = average_precisions(predictions, gts, num_classes, iou_threshold, voc=voc)
    if verbose:
        logging.info('Mean Average Precision results:')
        for class_index in range(num_classes):
            if not torch.isnan(ap[class_index]) or _has_positive(gts, class_index):
                logging.info(f'{class_labels[class_index]}: {ap[class_index].item():6f}')
        logging.info(f'Total mean: {mAP:6f}')
    return mAP


def _has_positive(gts, class_index):
    for g in gts:
        if g.numel() == 0:
            continue
        sel = g[:, 4].long() == class_index
        if g.size(1) > DIFFICULT_INDEX:
            sel = sel & (g[:, DIFFICULT_INDEX] == 0)
        if bool(sel.any()):
            return True
    return False
