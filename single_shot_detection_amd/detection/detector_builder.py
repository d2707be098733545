"""Detector builder -- mirror of detection/detector_builder.py:12-150."""
import functools

import torch
import torch.nn as nn

from ..bf.modules import conv
from ..bf.modules import features as _features
from . import anchor_generators as _anchor_generators
from .detector import Detector
from .modules import predictors


def build(base, anchor_generator_params, num_classes, features, use_depthwise=False, extras={}, predictor={}, heads={}):
    extra_layers = extras.get('layers', [])
    Features = getattr(_features, features['name'])
    features = Features(base, use_depthwise=use_depthwise, **features)
    num_scales = features.num_outputs + len(extra_layers)
    source_out_channels = features.get_out_channels()
    anchor_generator_builder = getattr(_anchor_generators, anchor_generator_params['type']).build_anchor_generators
    anchor_generators = anchor_generator_builder(**anchor_generator_params)
    assert num_scales == len(anchor_generators)
    num_boxes = [x.num_boxes for x in anchor_generators]
    extras = get_extras(source_out_channels, use_depthwise=use_depthwise, **extras)
    predictor = get_predictor(source_out_channels, num_boxes, num_classes, use_depthwise, predictor_args=predictor)
    out_channels = predictor.out_channels if predictor else source_out_channels
    heads = get_heads(out_channels, num_boxes, num_classes, **heads)
    return Detector(features, extras, predictor, heads, num_classes, anchor_generators=anchor_generators)


def get_extras(source_out_channels, use_depthwise=False, layers=(), activation={'name': 'ReLU', 'args': {'inplace': True}},
               initializer={'name': 'xavier_normal_'}, batch_norm={}):
    """detector_builder.py:57-109: 's' = 1x1 conv-BN-ReLU to Cout/2 then 3x3 stride-2 conv-BN-ReLU to Cout;
    '' = the same with a 3x3 valid conv; 'm' = 3x3/2 max-pool."""
    extras = nn.ModuleList()
    in_channels = source_out_channels[-1]
    for type_, out_channels in layers:
        blocks = []
        if type_ == 'm':
            out_channels = in_channels
            blocks.append(nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        elif type_ in ('s', ''):
            blocks.append(conv.Conv2dBn(in_channels, out_channels // 2, kernel_size=1, bias=False, activation_params=activation,
                                        use_bn=True, batch_norm_params=batch_norm))
            in_channels = out_channels // 2
            kw = dict(kernel_size=3, bias=False, activation_params=activation, use_bn=True, batch_norm_params=batch_norm)
            if type_ == 's':
                kw.update(stride=2, padding=1)
            block = conv.DepthwiseConv2dBn if use_depthwise else conv.Conv2dBn
            blocks.append(block(in_channels, out_channels, **kw))
        else:
            raise ValueError(f'Unknown layer type: {type_}')
        source_out_channels.append(out_channels)
        extras.append(nn.Sequential(*blocks))
        in_channels = out_channels
    initializer_ = functools.partial(getattr(nn.init, initializer['name']), **initializer.get('args', {}))

    def _init_extras(layer):
        if isinstance(layer, nn.Conv2d):
            initializer_(layer.weight)
            layer.bias is not None and nn.init.zeros_(layer.bias)
    extras.apply(_init_extras)
    return extras


def get_heads(out_channels, num_boxes, num_classes, initializer={'name': 'normal_', 'args': {'mean': 0, 'std': 0.01}},
              score_head_bias_init=0.0):
    """detector_builder.py:111-137.  The parameters are ordinary nn.Conv2d weights/biases (state_dict keys
    ``heads.<i>.score|loc.weight|bias``) kept in channels_last memory = the [n][ky][kx][cin] rows the GEMM reads."""
    initializer_ = functools.partial(getattr(nn.init, initializer['name']), **initializer.get('args', {}))
    heads = nn.ModuleList()
    for in_channels, nb in zip(out_channels, num_boxes):
        score_head = nn.Conv2d(in_channels, nb * num_classes, kernel_size=3, padding=1, bias=True)
        loc_head = nn.Conv2d(in_channels, nb * 4, kernel_size=3, padding=1, bias=True)
        initializer_(score_head.weight)
        nn.init.constant_(score_head.bias, score_head_bias_init)
        initializer_(loc_head.weight)
        nn.init.zeros_(loc_head.bias)
        for m in (score_head, loc_head):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        heads.append(nn.ModuleDict({'score': score_head, 'loc': loc_head}))
    return heads


def get_predictor(source_out_channels, num_boxes, num_classes, use_depthwise, predictor_args):
    if not predictor_args:
        return None
    return predictors.SharedConvPredictor(source_out_channels, num_boxes, num_classes, use_depthwise, **predictor_args)
