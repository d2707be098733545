"""BoxCoder -- mirror of detection/box_coder.py:4-57 on libssdk (csrc/loss.hip)."""
import torch

from .. import _lib


class BoxCoder(torch.nn.Module):
    def __init__(self, xy_scale, wh_scale, eps=1e-8):
        super(BoxCoder, self).__init__()
        self.xy_scale = xy_scale
        self.wh_scale = wh_scale
        self.eps = eps

    @staticmethod
    def _shape(boxes, priors):
        A = priors.size(0)
        assert boxes.size(-1) == 4 and boxes.numel() % (A * 4) == 0
        return boxes.numel() // (A * 4), A

    def encode_box(self, boxes, priors, inplace=False):
        """boxes [Batch, AnchorBoxes, 4] centroid form -> encoded (box_coder.py:13-34).  ``inplace`` selects the
        reference's in-place arithmetic (eps added after the divide) and overwrites ``boxes``."""
        _lib.require_cuda(boxes, priors)
        B, A = self._shape(boxes, priors)
        src = boxes if boxes.is_contiguous() and boxes.dtype == torch.float32 else boxes.contiguous().float()
        out = src if inplace else torch.empty_like(src)
        _lib.check(_lib.lib().ssdk_encode_box(_lib.ptr(src), _lib.ptr(priors.contiguous().float()), _lib.ptr(out), B, A,
                                              float(self.xy_scale), float(self.wh_scale), float(self.eps),
                                              1 if inplace else 0, _lib.current_stream()), 'ssdk_encode_box')
        if inplace and out is not boxes:
            boxes.copy_(out)
            return boxes
        return out

    def decode_box(self, boxes, priors, inplace=torch.tensor(0)):
        """encoded [Batch, AnchorBoxes, 4] -> centroid boxes (box_coder.py:37-57)."""
        _lib.require_cuda(boxes, priors)
        B, A = self._shape(boxes, priors)
        do_inplace = bool(inplace)
        src = boxes if boxes.is_contiguous() and boxes.dtype == torch.float32 else boxes.contiguous().float()
        out = src if do_inplace else torch.empty_like(src)
        _lib.check(_lib.lib().ssdk_decode_box(_lib.ptr(src), _lib.ptr(priors.contiguous().float()), _lib.ptr(out), B, A,
                                              float(self.xy_scale), float(self.wh_scale), 1 if do_inplace else 0,
                                              _lib.current_stream()), 'ssdk_decode_box')
        if do_inplace and out is not boxes:
            boxes.copy_(out)
            return boxes
        return out
