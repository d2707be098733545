"""BatchContainer -- mirror of bf/core/batch_container.py:8-58 with a device-resident mixup.

Same constructor, ``to_``, ``mixup_``, ``pin_memory`` and ``get``.  Once the images are on the GPU (``to_``), ``mixup_`` runs on
libssdk: the random draws are made on the host with the same three calls in the same order as the reference
(``np.random.beta``, ``torch.randperm``, ``torch.rand``), so a seeded run mixes the same images with the same weights; the
blend and the ragged ground-truth concatenation happen on the device (csrc/metrics.hip) and the targets stay there as
views of one packed buffer."""
import numpy as np
import torch

import enum

from ... import _lib


class TargetTypes(enum.Enum):
    """What a batch carries besides the images (the reference keeps this enum in bf/core/target_types.py)."""
    Boxes = 'boxes'
    NoTarget = 'no_target'

SCORE_INDEX = 5   # bf/datasets/detection_dataset.py:14
GT_ROW = 6


class BatchContainer(object):
    def __init__(self, batch, target_type):
        imgs, targets = zip(*batch)
        self.imgs = torch.stack(imgs, dim=0)
        if target_type == TargetTypes.Boxes:
            self.targets = targets
        else:
            raise ValueError(f'Unknown type {target_type}')
        self.target_type = target_type

    def to_(self, device):
        self.imgs = self.imgs.to(device, non_blocking=True)
        return self

    def mixup_(self, alpha, p):
        lam = np.random.beta(alpha, alpha)            # batch_container.py:26-28: the same draws, in the same order
        index = torch.randperm(self.imgs.size(0))
        roll = torch.rand(self.imgs.size(0)) < p
        _lib.require_cuda(self.imgs)                   # the device path is the only path: call to_(device) first (the reference's
                                                       # trainer does: callbacks.to_device runs before callbacks.mixup, main.py:91)
        lib = _lib.lib()
        dev = self.imgs.device
        B = self.imgs.size(0)
        imgs = self.imgs.contiguous().float()
        index_d = index.to(torch.int32).to(dev, non_blocking=True)
        roll_d = roll.to(torch.uint8).to(dev, non_blocking=True)
        out = torch.empty_like(imgs)
        _lib.check(lib.ssdk_mixup_images(_lib.ptr(imgs), _lib.ptr(out), B, imgs[0].numel(), _lib.ptr(index_d), _lib.ptr(roll_d), float(lam),
                                         _lib.current_stream()), 'ssdk_mixup_images')
        self.imgs = out
        if self.target_type == TargetTypes.Boxes:
            from ...detection.target_assigner import pack_ground_truth
            # every attribute column travels (the reference clones whole rows, :37-41): e.g. `difficult` at column 6, which
            # detection/metrics/mean_average_precision.py:22 reads when the rows are wider than 6
            widths = {int(t.size(1)) for t in self.targets if t.dim() == 2 and t.size(0) > 0}
            if len(widths) > 1:
                raise ValueError(f'mixup_: ground-truth rows of different widths in one batch: {sorted(widths)}')
            row = max(widths.pop() if widths else GT_ROW, GT_ROW)
            rows, offs, total = pack_ground_truth(list(self.targets), dev, row=row)
            rows_out = torch.empty((2 * max(total, 1), row), dtype=torch.float32, device=dev)
            offs_out = torch.empty((B + 1,), dtype=torch.int32, device=dev)
            _lib.check(lib.ssdk_mixup_ground_truth(_lib.ptr(rows), row, _lib.ptr(offs), B, _lib.ptr(index_d), _lib.ptr(roll_d), float(lam),
                                                   _lib.ptr(rows_out), _lib.ptr(offs_out), _lib.current_stream()), 'ssdk_mixup_ground_truth')
            # per-image views of the packed buffer; the sizes are known on the host (no D2H sync)
            counts = [int(t.size(0)) if t.dim() == 2 else 0 for t in self.targets]
            sizes = [counts[i] + (counts[int(index[i])] if bool(roll[i]) else 0) for i in range(B)]
            self.targets = list(torch.split(rows_out[:sum(sizes)], sizes, dim=0))
        return self

    def pin_memory(self):
        self.imgs = self.imgs.pin_memory()
        if self.target_type == TargetTypes.Boxes:
            self.targets = [t.pin_memory() for t in self.targets]
        return self

    def get(self):
        return self.imgs, self.targets
