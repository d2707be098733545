"""detection/matcher.py constants.  The matcher itself (match_per_prediction, matcher.py:33-56) is fused into
``ssdk_encode_ground_truth`` (csrc/match.hip); ``match_per_prediction`` below exposes it for a given box set."""
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
