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


def ssd_box_sizes(aspect_ratios, num_branches, img_size, min_scale=None, max_scale=None, min_size=None, max_size=None):
    """(w, h) table of SsdAnchorGenerator._generate_anchors (ssd.py:120-136), float32 [num_ratios * num_branches, 2], with the
    reference's mix of fp32 tensor and python-float arithmetic: sizes = linspace(...) in fp32 (times the image size in fp32 when
    scales are given); w = size * sqrt(r), h = size / sqrt(r) with the square root rounded to fp32 first (a python scalar joins an
    fp32 tensor op as fp32); the extra box sqrt(min * max) with the product in fp32 and the root in double."""
    import math
    img_w, img_h = img_size
    lin = np.empty(num_branches + 1, np.float32)
    if min_size is not None and max_size is not None:
        _lib.check(_lib.lib().ssdk_linspace_f32(float(min_size), float(max_size), num_branches + 1, lin.ctypes.data_as(C.c_void_p)), 'ssdk_linspace_f32')
        sizes = np.stack([lin, lin], 1)                                              # ssd.py:102 .expand(-1, 2)
    else:
        _lib.check(_lib.lib().ssdk_linspace_f32(float(min_scale), float(max_scale), num_branches + 1, lin.ctypes.data_as(C.c_void_p)), 'ssdk_linspace_f32')
        sizes = np.stack([lin * np.float32(img_w), lin * np.float32(img_h)], 1)      # ssd.py:125
    nr = len(aspect_ratios) + 1
    hws = np.empty((nr * num_branches, 2), np.float32)
    for j in range(num_branches):
        mn, mx = sizes[j], sizes[j + 1]
        for i, r in enumerate(aspect_ratios):
            sr = np.float32(math.sqrt(r))
            hws[j * nr + i, 0] = mn[0] * sr                                          # :132
            hws[j * nr + i, 1] = mn[1] / sr                                          # :133
        hws[j * nr + nr - 1, 0] = np.float32(math.sqrt(float(np.float32(mn[0] * mx[0]))))   # :135
        hws[j * nr + nr - 1, 1] = np.float32(math.sqrt(float(np.float32(mn[1] * mx[1]))))   # :136
    return hws


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
        if (min_scale is None) == (min_size is None):
            raise ValueError('one of "min_scale" / "min_size" is required')
        if (max_scale is None) if min_scale is not None else (max_size is None):
            # without a maximum the reference writes the extra box one slot past the ratios of a branch (ssd.py:135 uses the loop
            # variable i after the loop): an IndexError for the last branch -- no sample config does this
            raise NotImplementedError('SsdAnchorGenerator without max_scale / max_size fails in the reference itself (ssd.py:135-136)')
        self.min_scale = None if min_scale is None else float(np.float32(min_scale))   # (a 0-dim fp32 tensor from the builder, ssd.py:43-44)
        self.max_scale = None if max_scale is None else float(np.float32(max_scale))
        self.min_size, self.max_size = min_size, max_size
        self.num_branches = num_branches
        self.clip = clip  # the reference's clip branch is a no-op on a copy (ssd.py:147-149)
        self.offset = offset
        self.step = step
        self.aspect_ratios = []
        for ar in aspect_ratios:
            assert ar >= 1.0 or not flip
            self.aspect_ratios.append(ar)
            if ar > 1.0 and flip:
                self.aspect_ratios.append(1.0 / ar)
        self.num_ratios = len(self.aspect_ratios) + 1
        self.num_boxes = self.num_ratios * num_branches

    def _box_sizes(self, img_size):
        return ssd_box_sizes(self.aspect_ratios, self.num_branches, img_size, self.min_scale, self.max_scale, self.min_size, self.max_size)

    def _grid(self, img_size, feature_map_size):
        step_w = self.step if self.step is not None else img_size[0] / feature_map_size[0]   # ssd.py:111-118
        step_h = self.step if self.step is not None else img_size[1] / feature_map_size[1]
        return step_w, step_h, self.offset[0], self.offset[1]
