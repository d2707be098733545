"""SSD anchors -- mirror of detection/anchor_generators/ssd.py:12-151."""
import ctypes as C
import logging

import numpy as np

from ... import _lib
from ...utils import filter_kwargs
from ._anchor_generator import _AnchorGenerator


@filter_kwargs
def build_anchor_generators(num_scales=6,
                            sizes=None,
                            min_scale=None,
                            max_scale=None,
                            aspect_ratios=[[1.0, 2.0]] + [[1.0, 2.0, 3.0]] * 3 + [[1.0, 2.0]] * 2,
                            steps=None,
                            offsets=[0.5, 0.5],
                            num_branches=None):
    """ssd.py:12-53.  As in the reference, ``offsets`` is accepted but not forwarded (ssd.py:18,52)."""
    assert sizes is not None or (min_scale is not None and max_scale is not None)
    if steps is None:
        steps = [None] * num_scales
    else:
        assert len(steps) == num_scales
    if num_branches is None:
        num_branches = [1] * num_scales
    else:
        assert len(num_branches) == num_scales
    if min_scale is not None and max_scale is not None:
        scales = np.empty(num_scales + 1, np.float32)  # torch.linspace(min, max, L + 1), ssd.py:33
        _lib.check(_lib.lib().ssdk_linspace_f32(float(min_scale), float(max_scale), num_scales + 1,
                                                scales.ctypes.data_as(C.c_void_p)), 'ssdk_linspace_f32')
        logging.info(f'Detector (Scales: {scales[:-1]})')
    else:
        scales = None
    assert len(aspect_ratios) == num_scales
    anchor_generators = []
    for i, (ratios, step, branches) in enumerate(zip(aspect_ratios, steps, num_branches)):
        if scales is not None:
            kwargs = {'min_scale': scales[i], 'max_scale': scales[i + 1]}
        else:
            kwargs = {'min_size': sizes[i], 'max_size': sizes[i + 1]}
        anchor_generators.append(SsdAnchorGenerator(ratios, step=step, num_branches=branches, **kwargs))
    return anchor_generators


class SsdAnchorGenerator(_AnchorGenerator):
    def __init__(self, aspect_ratios, min_scale=None, max_scale=None, min_size=None, max_size=None, step=None,
                 offset=[.5, .5], num_branches=1, flip=True, clip=False):
        super(SsdAnchorGenerator, self).__init__()
        if max_scale is not None and min_scale is None:
            raise ValueError('"max_scale" should be provided along with "min_scale"')
        if max_size is not None and min_size is None:
            raise ValueError('"max_size" should be provided along with "min_size"')
        if min_scale is not None and min_size is not None:
            raise ValueError('Either "min_scale" or "min_size" should be provided')
        # The sample configs on the hot path use scales, one branch, the default offset/step, flip and no clip
        # (SURVEY.md §8a A1); the other constructor modes are outside round-1 scope and fail loudly.
        if min_size is not None or num_branches != 1 or step is not None or list(offset) != [.5, .5] or not flip:
            raise NotImplementedError('SsdAnchorGenerator: only min_scale/max_scale, num_branches=1, step=None, '
                                      'offset=[.5,.5], flip=True are implemented on the GPU path')
        if max_scale is None:
            raise NotImplementedError('SsdAnchorGenerator: max_scale is required on the GPU path')
        self.min_scale = np.float32(min_scale)
        self.max_scale = np.float32(max_scale)
        self.num_branches = num_branches
        self.clip = clip  # the reference's clip branch is a no-op on a copy (ssd.py:147-149)
        self.offset = offset
        self.step = step
        self.base_ratios = [float(r) for r in aspect_ratios]
        self.aspect_ratios = []
        for ar in aspect_ratios:
            assert ar >= 1.0 or not flip
            self.aspect_ratios.append(ar)
            if ar > 1.0 and flip:
                self.aspect_ratios.append(1.0 / ar)
        self.num_ratios = len(self.aspect_ratios) + 1
        self.num_boxes = self.num_ratios * num_branches

    def _box_sizes(self, img_size):
        img_w, img_h = img_size
        ratios = np.asarray(self.base_ratios, dtype=np.float64)
        hws = np.empty((self.num_boxes, 2), np.float32)
        n = _lib.lib().ssdk_anchor_sizes_ssd(ratios.ctypes.data_as(C.c_void_p), len(ratios), float(self.min_scale),
                                             float(self.max_scale), int(img_w), int(img_h),
                                             hws.ctypes.data_as(C.c_void_p), self.num_boxes)
        if n != self.num_boxes:
            _lib.check(n if n < 0 else -1, 'ssdk_anchor_sizes_ssd')
        return hws
