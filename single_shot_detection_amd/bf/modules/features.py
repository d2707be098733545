"""Backbone tap -- restatement of bf/modules/features.py:18-49 (``Features``).  The backbone and neck stay stock
PyTorch-ROCm modules (north_star); this class only hands their intermediate maps to the hot path.  Maps are
produced in channels_last memory so the head GEMMs read them without a layout copy."""
import functools
import itertools

import torch
import torch.nn as nn

from ... import ops
from ...utils import filter_kwargs
from . import conv


def _init_layer(layer, initializer_):
    if isinstance(layer, nn.Conv2d):
        initializer_(layer.weight)
        layer.bias is not None and nn.init.zeros_(layer.bias)


def get_multiple_outputs(model, input_, output_layers):
    """bf/utils/torch_utils.py:7-37 for integer taps."""
    assert isinstance(model, nn.Sequential)
    x = input_
    idx = 0
    outputs = []
    for i, layer in enumerate(model):
        x = layer(x)
        if idx < len(output_layers) and i == output_layers[idx]:
            outputs.append(x)
            idx += 1
    return outputs, x


class Features(nn.Module):
    @filter_kwargs
    def __init__(self, base, out_layers, last_feature_layer=None, initializer={'name': 'xavier_normal_'}):
        super(Features, self).__init__()
        assert isinstance(base.features, nn.Sequential)
        feature_layers = list(base.features.children())
        if last_feature_layer is not None:
            feature_layers = feature_layers[:(last_feature_layer + 1)]
        self.base = nn.Sequential(*feature_layers)
        self.out_layers = out_layers
        self.num_outputs = len(out_layers)
        initializer_ = functools.partial(getattr(nn.init, initializer['name']), **initializer.get('args', {}))
        self.init_layer = functools.partial(_init_layer, initializer_=initializer_)

    def forward(self, x):
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        return get_multiple_outputs(self.base, x, self.out_layers)

    def get_out_channels(self):
        was_training = self.training
        self.eval()
        p = next(self.parameters())
        dummy = torch.ones((1, 3, 300, 300), dtype=p.dtype, device=p.device)
        with torch.no_grad():
            sources, _ = get_multiple_outputs(self.base, dummy, self.out_layers)
        self.train(was_training)
        return [s.size(1) for s in sources]


class FeaturePyramid(Features):
    """FPN neck -- restatement of bf/modules/features.py:52-120 (ref. arXiv:1612.03144), SURVEY.md §8f1.

    Lateral 1x1 convs and the 3x3 output blocks run on libssdk's implicit-GEMM kernels (``ops.conv2d`` / ``Conv2dBn``), the
    top-down ``features[i] += interpolate(features[i+1], nearest)`` on ``ssdk_upsample_nearest_add``.  Module names
    (``pyramid_lateral``, ``pyramid_output``) follow the reference so checkpoints map 1:1."""

    def __init__(self, base, out_layers, pyramid_layers, pyramid_channels, interpolation_mode='nearest', use_depthwise=False,
                 activation={'name': 'ReLU', 'args': {'inplace': True}}, initializer={'name': 'xavier_normal_'}, **kwargs):
        super(FeaturePyramid, self).__init__(base, out_layers, initializer=initializer, **kwargs)
        assert pyramid_layers >= len(out_layers)
        if interpolation_mode != 'nearest':
            raise NotImplementedError("FeaturePyramid: only interpolation_mode='nearest' is on the GPU path")
        self.pyramid_layers = pyramid_layers
        self.pyramid_channels = pyramid_channels
        self.interpolation_mode = interpolation_mode
        self.use_depthwise = use_depthwise
        self.num_outputs = pyramid_layers
        self.pyramid_lateral = nn.ModuleList()
        self.pyramid_output = nn.ModuleList()
        base_out_channels = super(FeaturePyramid, self).get_out_channels()
        conv_op = functools.partial(conv.Conv2dBn, groups=pyramid_channels) if use_depthwise else conv.Conv2dBn
        for in_channels in base_out_channels:
            lateral = nn.Conv2d(in_channels, pyramid_channels, kernel_size=1)
            self.pyramid_lateral.append(lateral)
            self.pyramid_output.append(conv_op(pyramid_channels, pyramid_channels, kernel_size=3, padding=1, activation_params=activation))
        for _ in range(pyramid_layers - len(base_out_channels)):
            self.pyramid_output.append(conv_op(pyramid_channels, pyramid_channels, kernel_size=3, padding=1, stride=2,
                                               activation_params=activation))
        self.pyramid_lateral.apply(self.init_layer)
        self.pyramid_output.apply(self.init_layer)
        for m in self.pyramid_output:   # init_layer rewrote the weights: restore the channels_last memory the GEMM reads
            m.conv.weight.data = m.conv.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        sources, _ = super(FeaturePyramid, self).forward(x)
        features = [ops.conv2d(s, lat.weight, lat.bias) for s, lat in zip(sources, self.pyramid_lateral)]   # features.py:104
        for i in reversed(range(len(features) - 1)):                                                       # :106-107
            features[i] = ops.upsample_add(features[i], features[i + 1])
        outputs = []
        for output_layer, feature in itertools.zip_longest(self.pyramid_output, features):                 # :109-115
            outputs.append(output_layer(feature if feature is not None else outputs[-1]))
        return outputs, outputs[-1]

    def get_out_channels(self):
        return [self.pyramid_channels] * self.pyramid_layers
