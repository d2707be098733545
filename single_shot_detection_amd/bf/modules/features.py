"""Backbone tap -- restatement of bf/modules/features.py:18-49 (``Features``).  The backbone and neck stay stock
PyTorch-ROCm modules (north_star); this class only hands their intermediate maps to the hot path.  Maps are
produced in channels_last memory so the head GEMMs read them without a layout copy."""
import functools

import torch
import torch.nn as nn

from ...utils import filter_kwargs


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
