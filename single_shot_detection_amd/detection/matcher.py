"""detection/matcher.py.  In the training path the matcher (match_per_prediction, matcher.py:33-56) is fused with the IoU into
``ssdk_encode_ground_truth`` (csrc/match.hip) and no [Boxes, AnchorBoxes] matrix exists; ``match_per_prediction`` below is the
reference's own function on a given weight matrix (``ssdk_match_per_prediction``), ``match_boxes`` the fused form for one box set.
``match_bipartite`` (matcher.py:7-31) is not called by anything in the reference and is not reproduced."""
import torch

from .. import _lib

NOT_MATCHED = -2  # matcher.py:4
IGNORE = -1       # matcher.py:5


def match_boxes(gt_boxes, anchors, matched_threshold, unmatched_threshold=None):
    """box_idx int64 [A] for one image: IoU (box_utils.py:83-101) + match_per_prediction (matcher.py:33-56,
    force_match_for_each_target=True) on the GPU.  ``gt_boxes`` [G, >=4] corner form, ``anchors`` [A,4] centroid."""
    from .target_assigner import TargetAssigner
    if unmatched_threshold is None:
        unmatched_threshold = matched_threshold
    gt = torch.zeros((gt_boxes.size(0), 6), dtype=torch.float32, device=gt_boxes.device)
    gt[:, :4] = gt_boxes[:, :4]
    _, idx = TargetAssigner(matched_threshold, unmatched_threshold).encode_ground_truth([gt], anchors, return_box_idx=True)
    return idx[0].long()


def match_per_prediction(weights, matched_threshold, unmatched_threshold=None, force_match_for_each_target=True):
    """
    Args:
        weights: torch.tensor(:shape [Boxes, AnchorBoxes]) on the GPU
    Returns:
        box_idx: torch.tensor(:shape [AnchorBoxes]) int64 -- matcher.py:33-56
    """
    if unmatched_threshold is None:
        unmatched_threshold = matched_threshold
    else:
        assert matched_threshold >= unmatched_threshold   # matcher.py:43
    _lib.require_cuda(weights)
    if weights.dim() != 2 or weights.size(0) == 0:
        raise ValueError('weights must be [Boxes, AnchorBoxes] with at least one box (torch.max over an empty dim fails in the reference)')
    lib = _lib.lib()
    w = weights.float().contiguous()
    G, A = w.shape
    box_idx = torch.empty((A,), dtype=torch.int64, device=w.device)
    ws = _lib.scratch(lib.ssdk_match_per_prediction_workspace_bytes(G), w.device, 'match_per_prediction')
    _lib.check(lib.ssdk_match_per_prediction(_lib.ptr(w), G, A, float(matched_threshold), float(unmatched_threshold),
                                             int(bool(force_match_for_each_target)), _lib.ptr(box_idx), _lib.ptr(ws), ws.numel(),
                                             _lib.current_stream()), 'ssdk_match_per_prediction')
    return box_idx
