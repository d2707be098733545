"""Postprocessor -- mirror of detection/postprocessor.py:9-78 on libssdk (csrc/postprocess.hip)."""
import torch

from .. import _lib


class Postprocessor(object):
    def __init__(self, box_coder, score_threshold, nms, score_converter='SOFTMAX', max_total=None):
        self.box_coder = box_coder
        self.score_threshold = score_threshold
        self.nms_args = dict(nms)  # {max_per_class, overlap_threshold[, soft, sigma]}  (box_utils.py:166)
        self.max_total = max_total
        self.score_converter = score_converter
        if score_converter not in ('SIGMOID', 'SOFTMAX'):
            raise ValueError(f'Wrong value for score_converter: {score_converter}')
        self.soft = bool(self.nms_args.get('soft', False))      # box_utils.py:166 `soft`, `sigma`
        self.sigma = float(self.nms_args.get('sigma', 0.5))
        self.last_nms_candidates = None

    def postprocess(self, prediction, priors):
        """
        Args:
            prediction: tuple of
                torch.tensor(:shape [Batch, AnchorBoxes * Classes])
                torch.tensor(:shape [Batch, AnchorBoxes * 4])
            priors: torch.tensor(:shape [AnchorBoxes, 4]
        Returns:
            processed: list(:len Batch) of torch.tensor(:shape [Boxes_i, 6] ~ {[0-3] - box, [4] - class, [5] - score})
        """
        out, counts = self.postprocess_padded(prediction, priors)
        counts = counts.tolist()  # the one D2H sync: result shapes are data-dependent
        return [out[i, :n] for i, n in enumerate(counts)]

    def postprocess_padded(self, prediction, priors):
        """Device-resident form: (out [Batch, cap, 6], counts int32 [Batch]) with no host synchronisation."""
        b_scores, b_boxes = prediction
        _lib.require_cuda(b_scores, b_boxes, priors)
        lib = _lib.lib()
        batch_size = b_scores.size(0)
        num_priors = priors.size(0)
        b_scores = b_scores.float().contiguous()   # postprocessor.py:39-40
        b_boxes = b_boxes.float().contiguous()
        priors = priors.float().contiguous()
        num_classes = b_scores.numel() // (batch_size * num_priors)
        softmax = 1 if self.score_converter == 'SOFTMAX' else 0
        ncls = num_classes - 1 if softmax else num_classes
        mpc = self.nms_args.get('max_per_class')
        max_per_class = int(mpc) if mpc is not None else 0          # box_utils.py:166 max_per_class=None: every candidate enters NMS
        max_total = int(self.max_total) if self.max_total is not None else 0
        cap = max_total if max_total > 0 else ncls * (min(max_per_class, num_priors) if max_per_class > 0 else num_priors)
        if cap * batch_size * 24 > (8 << 30):
            raise ValueError('postprocess without max_total and without max_per_class would return up to '
                             f'{cap} rows per image: set max_total')
        dev = b_scores.device
        need = lib.ssdk_postprocess_workspace_bytes_ex(batch_size, num_priors, num_classes, softmax, max_per_class, max_total, int(self.soft))
        ws = _lib.scratch(need, dev, 'postprocess')   # lives for this call only
        out = torch.empty((batch_size, cap, 6), dtype=torch.float32, device=dev)
        counts = torch.empty((batch_size,), dtype=torch.int32, device=dev)
        cand = torch.empty((batch_size,), dtype=torch.int64, device=dev)
        _lib.check(lib.ssdk_postprocess(_lib.ptr(b_scores), _lib.ptr(b_boxes), _lib.ptr(priors), batch_size, num_priors,
                                        num_classes, softmax, float(self.score_threshold), max_per_class,
                                        float(self.nms_args['overlap_threshold']), int(self.soft), self.sigma, max_total,
                                        float(self.box_coder.xy_scale), float(self.box_coder.wh_scale), _lib.ptr(out), cap,
                                        _lib.ptr(counts), _lib.ptr(cand), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                   'ssdk_postprocess')
        self.last_nms_candidates = cand
        return out, counts
