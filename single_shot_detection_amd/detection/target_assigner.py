"""TargetAssigner -- mirror of detection/target_assigner.py:17-63 on libssdk (csrc/match.hip)."""
import numpy as np
import torch

from .. import _lib

# detection/target_assigner.py:7-14
LOC_INDEX_START = 0
LOC_INDEX_END = 4
CLASS_INDEX = 4
SCORE_INDEX = 5
TARGET_SIZE = 6

NEGATIVE_CLASS = 0  # bf/datasets/detection_dataset.py:17
IGNORE_CLASS = -1

GT_ROW = 6  # x1, y1, x2, y2, class, score (bf/datasets/detection_dataset.py:11-15); extra columns are dropped


class PackedGroundTruth(object):
    """A batch's ground truth already on the device in the library's layout: ``rows`` [capacity, 6] fp32 (boxes of all images back to
    back, rows past ``offsets[-1]`` are padding), ``offsets`` int32 [B + 1].  ``TargetAssigner.encode_ground_truth`` takes it in place of
    the list of per-image tensors -- with no host-side packing and no H2D copy in the call, the training step can be captured in a HIP
    graph (graphs.py): keep ONE PackedGroundTruth of fixed capacity and ``update_`` it between replays."""

    def __init__(self, rows, offsets):
        assert rows.is_cuda and offsets.is_cuda and rows.dim() == 2 and rows.shape[1] == GT_ROW and rows.dtype == torch.float32
        assert offsets.dtype == torch.int32 and offsets.dim() == 1
        self.rows, self.offsets = rows.contiguous(), offsets.contiguous()

    @classmethod
    def from_list(cls, ground_truth, device, capacity=None):
        rows, offs, total = pack_ground_truth(ground_truth, device)
        cap = total if capacity is None else int(capacity)
        if cap < total:
            raise ValueError(f'PackedGroundTruth: {total} boxes do not fit the capacity of {cap}')
        buf = torch.zeros((max(cap, 1), GT_ROW), dtype=torch.float32, device=device)
        buf[:total] = rows
        return cls(buf, offs.clone())

    def update_(self, ground_truth):
        """New boxes into the same buffers (same batch size, at most the capacity): what a training loop does between graph replays."""
        rows, offs, total = pack_ground_truth(ground_truth, self.rows.device)
        if total > self.rows.shape[0] or offs.numel() != self.offsets.numel():
            raise ValueError('PackedGroundTruth.update_: batch size or capacity exceeded')
        self.rows[:total].copy_(rows, non_blocking=True)
        self.offsets.copy_(offs, non_blocking=True)
        return self

    def __len__(self):
        return self.offsets.numel() - 1


def pack_ground_truth(ground_truth, device, row=GT_ROW):
    """list[B] of [G_i, >=row] tensors (any device) -> (rows [sum G, row] fp32, offsets int32 [B+1]) on ``device``.

    One pinned staging buffer and one async H2D copy per batch (the reference moves each image's tensor
    separately, target_assigner.py:35).  ``row``: columns kept per box (6 for the target assigner; mixup keeps every
    attribute, e.g. the `difficult` flag at column 6 that the evaluation metric reads)."""
    GT_ROW = row   # (shadows the module constant below)
    counts = [int(g.size(0)) if g.dim() == 2 else 0 for g in ground_truth]
    total = sum(counts)
    if all((not g.is_cuda) for g in ground_truth):
        stage = torch.empty((total * GT_ROW + len(counts) + 1,), dtype=torch.float32)
        if torch.cuda.is_available():
            stage = stage.pin_memory()
        rows = stage[:total * GT_ROW].view(total, GT_ROW)
        offs = stage[total * GT_ROW:].view(torch.int32)
        pos = 0
        offs[0] = 0
        for i, (g, n) in enumerate(zip(ground_truth, counts)):
            if n:
                rows[pos:pos + n] = g[:, :GT_ROW].to(torch.float32)
            pos += n
            offs[i + 1] = pos
        dev = stage.to(device, non_blocking=True)
        return dev[:total * GT_ROW].view(total, GT_ROW), dev[total * GT_ROW:].view(torch.int32), total
    rows = torch.cat([g[:, :GT_ROW].to(device=device, dtype=torch.float32) for g, n in zip(ground_truth, counts) if n], dim=0) \
        if total else torch.zeros((0, GT_ROW), dtype=torch.float32, device=device)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32).to(device, non_blocking=True)
    return rows.contiguous(), offs, total


class TargetAssigner(object):
    def __init__(self, matched_threshold, unmatched_threshold):
        self.matched_threshold = matched_threshold
        self.unmatched_threshold = unmatched_threshold
        self._ws = None

    def encode_ground_truth(self, ground_truth, anchors, return_box_idx=False):
        """
        Args:
            ground_truth: list(:len Batch) of torch.tensor(:shape [Boxes_i, >=6])
            anchors: torch.tensor(:shape [AnchorBoxes, 4]) on the GPU
        Returns:
            target: torch.tensor(:shape [Batch, AnchorBoxes, 6]) on anchors.device
            (the reference runs this on the CPU because its anchors live there; here anchors are device-resident
            and so is the result -- the ``target.to(device)`` of detection/init.py:115 becomes a no-op)
        """
        _lib.require_cuda(anchors)
        lib = _lib.lib()
        device = anchors.device
        batch_size = len(ground_truth)
        num_anchors = anchors.size(0)
        anchors = anchors.contiguous().float()
        if isinstance(ground_truth, PackedGroundTruth):
            rows, offs, total = ground_truth.rows, ground_truth.offsets, ground_truth.rows.shape[0]   # (total = capacity: padding is skipped on the device)
        else:
            rows, offs, total = pack_ground_truth(ground_truth, device)
        target = torch.empty((batch_size, num_anchors, TARGET_SIZE), dtype=torch.float32, device=device)
        box_idx = torch.empty((batch_size, num_anchors), dtype=torch.int32, device=device) if return_box_idx else None
        need = lib.ssdk_encode_ground_truth_workspace_bytes(batch_size, total)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty((max(need, 4096),), dtype=torch.uint8, device=device)
        _lib.check(lib.ssdk_encode_ground_truth(_lib.ptr(rows), GT_ROW, _lib.ptr(offs), batch_size, total,
                                                _lib.ptr(anchors), num_anchors, float(self.matched_threshold),
                                                float(self.unmatched_threshold), _lib.ptr(target), _lib.ptr(box_idx),
                                                _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()),
                   'ssdk_encode_ground_truth')
        # keep the staging tensors alive until the stream has consumed them
        target._ssdk_keepalive = (rows, offs)
        return (target, box_idx) if return_box_idx else target
