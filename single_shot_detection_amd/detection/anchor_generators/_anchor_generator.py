"""Base anchor generator -- mirror of detection/anchor_generators/_anchor_generator.py:4-20.

The reference generates on the CPU (``device='cpu'`` default, never overridden) and the caller re-uploads every
step; here ``generate`` returns a cached DEVICE tensor written by ``ssdk_anchors_level`` (csrc/anchors.hip)."""
import ctypes as C

import numpy as np
import torch

from ... import _lib


class _AnchorGenerator(object):
    num_boxes = 0

    def _box_sizes(self, img_size):
        """host float32 [num_boxes, 2] (w, h) table for an image size (w, h)."""
        raise NotImplementedError

    def _grid(self, img_size, feature_map_size):
        """(step_w, step_h, offset_x, offset_y) of the centre grid as python floats (ssd.py:111-118, :138-139)."""
        return img_size[0] / feature_map_size[0], img_size[1] / feature_map_size[1], 0.5, 0.5

    def _generate_anchors(self, img_size, feature_map_size, device='cuda'):
        key = (tuple(img_size), tuple(feature_map_size), str(device))
        cache = self.__dict__.setdefault('_cache', {})
        if key not in cache:
            img_w, img_h = img_size
            layer_w, layer_h = feature_map_size
            hws = np.ascontiguousarray(self._box_sizes(img_size), dtype=np.float32)
            out = torch.empty((layer_h, layer_w, self.num_boxes, 4), dtype=torch.float32, device=device)
            _lib.require_cuda(out)
            step_w, step_h, off_x, off_y = self._grid(img_size, feature_map_size)
            _lib.check(_lib.lib().ssdk_anchors_level_ex(_lib.ptr(out), layer_h, layer_w, self.num_boxes,
                                                        hws.ctypes.data_as(C.c_void_p), float(step_w), float(step_h), float(off_x), float(off_y),
                                                        _lib.current_stream()), 'ssdk_anchors_level')
            cache[key] = out
        return cache[key]

    def generate(self, img, feature_map):
        """
        Args:
            img: torch.tensor(:shape [Batch, Channels, Height, Width])
            feature_map: torch.tensor(:shape [Batch, Channels, Height, Width])  (or a (H, W) pair)
        Returns:
            priors: torch.tensor(:shape [Height, Width, AspectRatios, 4]) on img.device
        """
        img_size = img.size(3), img.size(2)
        if isinstance(feature_map, (tuple, list)):
            feature_map_size = feature_map[1], feature_map[0]
        else:
            feature_map_size = feature_map.size(3), feature_map.size(2)
        device = img.device if img.is_cuda else torch.device('cuda')
        return self._generate_anchors(img_size, feature_map_size, device)
