"""SharedConvPredictor -- mirror of detection/modules/predictors.py:8-76 (RetinaNet tower: per layer one 3x3 conv
SHARED across levels, then ReLU, then a BatchNorm PER LEVEL; separate score and loc towers).  Module nesting and names
follow the reference so checkpoints map 1:1.  Each tower layer is ONE grouped implicit-GEMM launch over the five
levels (shared weights, ReLU fused into the epilogue, csrc/conv.hip) followed by the per-level BatchNorm kernels
(csrc/norm.hip)."""
import functools

import torch.nn as nn

from ... import ops
from ...bf.modules import conv


class SharedConvPredictor(nn.Module):
    def __init__(self, source_out_channels, num_boxes, num_classes, use_depthwise, num_layers=0, num_channels=256,
                 kernel_size=3, batch_norm={}, activation={'name': 'ReLU', 'args': {'inplace': True}},
                 initializer={'name': 'normal_', 'args': {'mean': 0, 'std': 0.01}}):
        super(SharedConvPredictor, self).__init__()
        if num_layers > 0:
            assert len(set(source_out_channels)) == 1
        self.convs = nn.ModuleDict()
        self.norms = nn.ModuleDict()
        for head in ['score', 'loc']:
            in_channels = source_out_channels[0]
            layers = nn.ModuleList()
            self.norms[head] = nn.ModuleList()
            for _ in range(num_layers):
                block = conv.DepthwiseConv2dBn if use_depthwise else conv.Conv2dBn
                layers.append(block(in_channels, num_channels, kernel_size=kernel_size, padding=1, bias=True,
                                    activation_params=None, use_bn=False))
                self.norms[head].append(nn.ModuleList([nn.BatchNorm2d(num_channels, **batch_norm) for _ in source_out_channels]))
                in_channels = num_channels
            self.convs[head] = layers
        self.activation = getattr(nn, activation['name'])(**activation.get('args', {}))
        self.out_channels = [num_channels] * len(source_out_channels)
        initializer_ = functools.partial(getattr(nn.init, initializer['name']), **initializer.get('args', {}))

        def _init_predictor(layer):
            if isinstance(layer, nn.Conv2d):
                initializer_(layer.weight)
                nn.init.zeros_(layer.bias)
        self.convs.apply(_init_predictor)

    def _layer(self, block, norms, xs):
        if isinstance(block, conv.Conv2dBn) and isinstance(self.activation, nn.ReLU) and block._hip_ok():
            c = block.conv
            if all(type(norm) is nn.BatchNorm2d and norm.training and ops.sync_group_of(norm) is None for norm in norms):
                # per-process statistics: taken in the shared convolution's epilogue, level by level (ops.conv2d_batch_norm)
                return ops.conv2d_batch_norm(list(xs), c.weight, c.bias, c.stride[0], c.padding[0], list(norms), conv_relu=True, bn_relu=False)
            ys = ops.conv2d(list(xs), c.weight, c.bias, stride=c.stride[0], padding=c.padding[0], relu=True)
            if all(type(norm) is nn.BatchNorm2d for norm in norms):
                # per-level norms; marked for synchronisation (detection.init(distributed=True)) the five share ONE all-reduce
                return ops.batch_norm_levels(ys, list(norms))
            return [ops.batch_norm(y, norm) if type(norm) is nn.BatchNorm2d else norm(y) for norm, y in zip(norms, ys)]
        return [norm(self.activation(block(x))) for norm, x in zip(norms, xs)]   # depthwise towers: stock ops

    def forward(self, sources):  # predictors.py:60-76: conv -> activation -> per-level norm
        score_sources = loc_sources = list(sources)
        if self.training:
            ops.prepare_weight_transposes(self.convs)   # (a shared tower weight is re-laid out once per step, not once per level and layer)
        for score_conv, loc_conv, score_norm, loc_norm in zip(self.convs['score'], self.convs['loc'], self.norms['score'],
                                                              self.norms['loc']):
            score_sources = self._layer(score_conv, score_norm, score_sources)
            loc_sources = self._layer(loc_conv, loc_norm, loc_sources)
        return score_sources, loc_sources
