"""RetinaNet anchors -- mirror of detection/anchor_generators/retina_net.py:10-54."""
import ctypes as C

import numpy as np

from ... import _lib
from ...utils import filter_kwargs
from ._anchor_generator import _AnchorGenerator


@filter_kwargs
def build_anchor_generators(aspect_ratios, min_level, max_level, scale, scales_per_level):
    return [RetinaAnchorGenerator(aspect_ratios, level, scale, scales_per_level) for level in range(min_level, max_level + 1)]


class RetinaAnchorGenerator(_AnchorGenerator):
    def __init__(self, aspect_ratios, level, scale, scales_per_level=1):
        self.aspect_ratios = aspect_ratios
        self.level = level
        self.scale = scale
        self.scales_per_level = scales_per_level
        self.num_boxes = len(aspect_ratios) * scales_per_level
        self.sizes = [scale * (2 ** (level + x / scales_per_level)) for x in range(scales_per_level)]

    def _box_sizes(self, img_size):
        ratios = np.asarray(self.aspect_ratios, dtype=np.float64)
        hws = np.empty((self.num_boxes, 2), np.float32)
        n = _lib.lib().ssdk_anchor_sizes_retina(ratios.ctypes.data_as(C.c_void_p), len(ratios), int(self.level),
                                                float(self.scale), int(self.scales_per_level),
                                                hws.ctypes.data_as(C.c_void_p), self.num_boxes)
        if n != self.num_boxes:
            _lib.check(n if n < 0 else -1, 'ssdk_anchor_sizes_retina')
        return hws
