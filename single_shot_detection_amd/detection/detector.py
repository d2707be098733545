"""Predictor / Detector -- mirror of detection/detector.py:8-96.

Same constructor arguments, attributes (``features``, ``extras``, ``predictor``, ``heads``, ``num_classes``,
``Detector.predictor``, ``Detector.priors``) and return layouts; the per-level head convolutions + NHWC flatten +
cat (detector.py:50-66) run as libssdk implicit GEMMs (modules/heads.py) and the anchors are device-resident
(anchor_generators/_anchor_generator.py) instead of being rebuilt on the CPU and re-uploaded every step.
"""
import torch
import torch.nn as nn

from .. import ops
from .modules.heads import multi_level_heads


class Predictor(nn.Module):
    def __init__(self, features, extras, predictor, heads, num_classes):
        super(Predictor, self).__init__()
        self.features = features
        self.extras = extras
        self.predictor = predictor
        self.heads = heads
        self.num_classes = num_classes

    def forward(self, img):
        """
        Args:
            img: torch.tensor(:shape [Batch, Channel, Height, Width])
        Returns:
            prediction: tuple of
                torch.tensor(:shape [Batch, AnchorBoxes * Classes])
                torch.tensor(:shape [Batch, AnchorBoxes * 4])
                list of the source maps the loc heads ran on (for anchor generation, detector.py:74)
        """
        sources, x = self.features(img)
        return self.forward_from_taps(list(sources), x)

    def forward_from_taps(self, sources, x):
        """Everything behind the backbone (detector.py:39-66): pyramid tail, optional predictor tower, heads.  ``sources``: the backbone's
        tapped maps, ``x``: its last map (the tail's input).  This is the part ``detection.init(graph_hot_path=True)`` captures."""
        sources = list(sources)
        if self.training and len(self.extras):
            ops.prepare_weight_transposes(self.extras)   # one re-layout launch for the whole tail's backward instead of one per layer
        for layer in self.extras:
            x = layer(x)
            sources.append(x)
        if self.predictor:
            score_sources, loc_sources = self.predictor(sources)
        else:
            score_sources = loc_sources = sources
        scores, locs = multi_level_heads(score_sources, loc_sources, self.heads)
        return scores, locs, loc_sources


class Detector(nn.Module):
    def __init__(self, *args, anchor_generators):
        super(Detector, self).__init__()
        self.predictor = Predictor(*args)
        self.priors = anchor_generators
        self._anchor_cache = {}

    def generate_anchors(self, img, sources):
        key = (tuple(img.shape[2:]), tuple(tuple(s.shape[2:]) for s in sources), str(img.device))
        if key not in self._anchor_cache:  # detector.py:82-86, once per geometry instead of once per step
            anchors = [g.generate(img, s).reshape(-1) for s, g in zip(sources, self.priors)]
            self._anchor_cache[key] = torch.cat(anchors, dim=0).view(-1, 4)
        return self._anchor_cache[key]

    def forward(self, img):
        scores, locs, locs_sources = self.predictor.forward(img)
        return scores, locs, self.generate_anchors(img, locs_sources)
