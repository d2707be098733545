"""Backbone tap -- restatement of bf/modules/features.py:18-49 (``Features``).  The backbone and neck stay stock
PyTorch-ROCm modules (north_star); this class only hands their intermediate maps to the hot path.  Maps are
produced in channels_last memory so the head GEMMs read them without a layout copy."""
import functools
import itertools

import torch
import torch.nn as nn
import torch.nn.functional as F

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
        if self.training:
            ops.prepare_weight_transposes(self.pyramid_lateral)
            ops.prepare_weight_transposes(self.pyramid_output)
        features = [ops.conv2d(s, lat.weight, lat.bias) for s, lat in zip(sources, self.pyramid_lateral)]   # features.py:104
        for i in reversed(range(len(features) - 1)):                                                       # :106-107
            features[i] = _upsample_add(features[i], features[i + 1], self.interpolation_mode)
        outputs = []
        for output_layer, feature in itertools.zip_longest(self.pyramid_output, features):                 # :109-115
            outputs.append(output_layer(feature if feature is not None else outputs[-1]))
        return outputs, outputs[-1]

    def get_out_channels(self):
        return [self.pyramid_channels] * self.pyramid_layers


def _upsample_add(fine, coarse, mode):
    """fine + F.interpolate(coarse, size of fine, mode) (features.py:107-108, :264-265).  'nearest' -- the default and what every sample
    config uses -- is the libssdk kernel; any other mode is torch's own interpolation (the neck is PyTorch-ROCm territory, SURVEY.md 2
    row 16), said once."""
    if mode == 'nearest':
        return ops.upsample_add(fine, coarse)
    conv._warn_stock('interpolate', f"interpolation_mode={mode!r}")
    return fine + F.interpolate(coarse, size=fine.shape[2:], mode=mode)


def update_existing(dict1, dict2):
    """bf/utils/misc_utils.py:31-34: fill in keys that are missing."""
    for k, v in dict2.items():
        if k not in dict1:
            dict1[k] = v


class ThinnedUshapeModule(nn.Module):
    """M2Det TUM -- restatement of bf/modules/features.py:215-270 (encoder of stride-2 3x3 blocks, decoder of 1x1 blocks with
    nearest upsample + skip add, 1x1 smoothing of every decoder stage).  Conv2dBn blocks and the upsample-add run on libssdk."""

    def __init__(self, in_channels, inner_channels, out_channels, num_scales, interpolation_mode='nearest', use_depthwise=False,
                 activation={'name': 'ReLU', 'args': {'inplace': True}}, initializer={'name': 'xavier_normal_'}):
        super(ThinnedUshapeModule, self).__init__()
        self.interpolation_mode = interpolation_mode
        self.down_layers = nn.ModuleList()
        self.up_layers = nn.ModuleList()
        self.smooth_layers = nn.ModuleList()
        conv_op = conv.DepthwiseConv2dBn if use_depthwise else conv.Conv2dBn
        for i in range(num_scales):
            if i > 0:
                self.down_layers.append(conv_op(in_channels if i == 1 else inner_channels, inner_channels, kernel_size=3, stride=2,
                                                padding=1, activation_params=activation))
                self.up_layers.append(conv_op(inner_channels, in_channels if i == 1 else inner_channels, kernel_size=1,
                                              activation_params=activation))
            self.smooth_layers.append(conv_op(in_channels if i == 0 else inner_channels, out_channels, kernel_size=1,
                                              activation_params=activation))

    def forward(self, x):
        down_path = [x]
        for layer in self.down_layers:
            x = layer(x)
            down_path.append(x)
        up_path = [x]
        for down_x, layer in zip(reversed(down_path[:-1]), reversed(self.up_layers)):
            x = _upsample_add(down_x, layer(x), self.interpolation_mode)   # features.py:263-265: interpolate to the skip's size, add the skip
            up_path.append(x)
        smooth = list(reversed(self.smooth_layers))
        if len(smooth) <= 8 and all(isinstance(l, conv.Conv2dBn) and l._hip_ok() and x.is_cuda for l, x in zip(smooth, up_path)):
            return ops.conv2d_bn_group(up_path, smooth)   # features.py:267: the scale branches are independent -> ONE grouped launch
        return [layer(x) for layer, x in zip(smooth, up_path)]


class ScalewiseFeatureAggregationModule(nn.Module):
    """M2Det SFAM -- restatement of bf/modules/features.py:273-300 (a squeeze-excite gate per scale)."""

    def __init__(self, num_channels, num_scales, reduction_ratio=16):
        super(ScalewiseFeatureAggregationModule, self).__init__()
        self.fc1 = nn.ModuleList()
        self.fc2 = nn.ModuleList()
        for _ in range(num_scales):
            self.fc1.append(nn.Conv2d(num_channels, num_channels // reduction_ratio, kernel_size=1))
            self.fc2.append(nn.Conv2d(num_channels // reduction_ratio, num_channels, kernel_size=1))

    def forward(self, features):
        assert len(features) == len(self.fc1)
        result = []
        if len(features) <= 8 and all(f.is_cuda for f in features):
            # the per-scale gates are independent (features.py:290-296): fc1 of every scale in one grouped launch, fc2 likewise
            pooled = [ops.global_avg_pool(f) for f in features]
            hidden = ops.conv2d_group(pooled, list(self.fc1), relu=True)    # fc1 + F.relu
            z = ops.conv2d_group(hidden, list(self.fc2))
            return [ops.sigmoid_gate(f, zz) for f, zz in zip(features, z)]   # feature * sigmoid(x)
        for feature, fc1, fc2 in zip(features, self.fc1, self.fc2):
            x = ops.global_avg_pool(feature)
            x = ops.conv2d(x, fc1.weight, fc1.bias, relu=True)     # fc1 + F.relu
            x = ops.conv2d(x, fc2.weight, fc2.bias)
            result.append(ops.sigmoid_gate(feature, x))             # feature * sigmoid(x)
        return result

    def forward_pieces(self, pieces):
        """forward([torch.cat(p, dim=1) for p in pieces]) for per-scale LISTS of maps (the TUM outputs features.py:385 concatenates): on
        libssdk the concatenated maps are never built (ops.sfam_pieces); anything the kernels do not take goes through torch.cat and
        forward().  SSDK_SFAM_CAT=1 forces that path (measurement / tests)."""
        import os
        assert len(pieces) == len(self.fc1)
        if not os.environ.get('SSDK_SFAM_CAT') and ops.sfam_pieces_ok(pieces, list(self.fc1), list(self.fc2)):
            return ops.sfam_pieces(pieces, list(self.fc1), list(self.fc2))
        return self.forward([torch.cat(p, dim=1) for p in pieces])


class MultilevelFeaturePyramid(Features):
    """M2Det MLFPN neck -- restatement of bf/modules/features.py:303-393."""

    def __init__(self, base, out_layers, num_scales, num_tums, base_reduced_channels=[256, 512], reduced_channels=128,
                 interpolation_mode='nearest', use_depthwise=False, activation={'name': 'ReLU', 'args': {'inplace': True}},
                 initializer={'name': 'xavier_normal_'}, tum={'inner_channels': 256, 'out_channels': 128},
                 sfam={'reduction_ratio': 16}, **kwargs):
        super(MultilevelFeaturePyramid, self).__init__(base, out_layers, initializer=initializer, **kwargs)
        assert len(out_layers) == len(base_reduced_channels)
        assert num_tums > 0
        self.num_outputs = num_scales
        self.num_tums = num_tums
        self.interpolation_mode = interpolation_mode
        self.base_reducers = nn.ModuleList()
        base_out_channels = super(MultilevelFeaturePyramid, self).get_out_channels()
        for in_channels, out_channels in zip(base_out_channels, base_reduced_channels):
            self.base_reducers.append(conv.Conv2dBn(in_channels, out_channels, kernel_size=1, activation_params=activation))
        tum = dict(tum)
        tum.update({'num_scales': num_scales})
        update_existing(tum, {'interpolation_mode': interpolation_mode, 'use_depthwise': use_depthwise, 'activation': activation})
        self.tum_out_channels = tum['out_channels']
        self.tums = nn.ModuleList()
        self.reducers = nn.ModuleList()
        self.tums.append(ThinnedUshapeModule(in_channels=sum(base_reduced_channels), **tum))
        for _ in range(1, num_tums):
            self.tums.append(ThinnedUshapeModule(in_channels=reduced_channels + self.tum_out_channels, **tum))
            self.reducers.append(conv.Conv2dBn(sum(base_reduced_channels), reduced_channels, kernel_size=1, activation_params=activation))
        sfam = dict(sfam)
        sfam.update({'num_channels': self.tum_out_channels * self.num_tums, 'num_scales': num_scales})
        self.sfam = ScalewiseFeatureAggregationModule(**sfam)
        for group in (self.base_reducers, self.tums, self.reducers, self.sfam):
            group.apply(self.init_layer)
        for m in self.modules():   # init rewrote the weights: restore the channels_last memory the GEMM kernels read
            if isinstance(m, conv.Conv2dBn):
                m.conv.weight.data = m.conv.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        sources, _ = super(MultilevelFeaturePyramid, self).forward(x)
        return self.neck(sources)

    def neck(self, sources):
        """Everything behind the backbone taps (features.py:363-393); separate so that a caller that already holds the taps
        (bench.py) can drive the libssdk part alone."""
        if self.training:   # one re-layout launch for the backward-data GEMMs of the ~90 convolutions below instead of one each
            for group in (self.base_reducers, self.tums, self.reducers, self.sfam):
                ops.prepare_weight_transposes(group)
        base_reduced = [reducer(source) for reducer, source in zip(self.base_reducers, sources)]
        size = base_reduced[0].shape[2:]
        upscaled = [base_reduced[0]] + [ops.upsample_nearest(f, size) for f in base_reduced[1:]]   # features.py:369-371
        base_features = torch.cat(upscaled, dim=1)
        features = [[f] for f in self.tums[0](base_features)]
        reduced_all = self._reducers_merged(base_features)
        for k, (tum, reducer) in enumerate(zip(self.tums[1:], self.reducers)):
            reduced = reduced_all[k] if reduced_all is not None else reducer(base_features)
            x = torch.cat([features[-1][-1], reduced], dim=1)                                      # :378-380
            for i, feature in enumerate(tum(x)):
                features[i].append(feature)
        features = self.sfam.forward_pieces(list(reversed(features)))                              # :385 torch.cat per scale + :387 sfam
        return features, features[-1]

    def _reducers_merged(self, base_features):
        """[reducer(base_features) for reducer in self.reducers] (features.py:379) as ONE convolution: the num_tums - 1 reducers are 1 x 1
        Conv2dBn blocks on the SAME input, so their weights stack along the output channels, and BatchNorm being per channel, one norm
        over the stacked channels is the seven norms.  The 200 MB base map (batch 16) is then read once instead of seven times, and the
        backward pass has one data-gradient GEMM instead of seven plus six full-size gradient additions.  The modules keep their own
        parameters and buffers (state_dict unchanged): gradients reach them through torch.cat, the running statistics are copied back.
        None when the blocks are not plain libssdk Conv2dBn blocks with identical settings (then they run one by one)."""
        rs = list(self.reducers)
        if len(rs) < 2 or not base_features.is_cuda:
            return None
        ok = all(isinstance(r, conv.Conv2dBn) and r._hip_ok() and type(r._modules.get('bn')) is nn.BatchNorm2d and r.conv.bias is None
                 and r.conv.kernel_size == (1, 1) and r.conv.stride == (1, 1) and r.conv.padding == (0, 0) and ops.sync_group_of(r.bn) is None
                 and r.bn.affine and r.bn.track_running_stats and r.bn.momentum == rs[0].bn.momentum and r.bn.eps == rs[0].bn.eps
                 and r.bn.training == rs[0].bn.training and r.conv.out_channels == rs[0].conv.out_channels for r in rs)
        if not ok:
            return None
        has_act = 'activation' in rs[0]._modules
        w = torch.cat([r.conv.weight for r in rs], dim=0).contiguous(memory_format=torch.channels_last)
        y = ops.conv2d(base_features, w, None, 1, 0)
        bn0 = rs[0].bn
        gamma, beta = torch.cat([r.bn.weight for r in rs]), torch.cat([r.bn.bias for r in rs])
        with torch.no_grad():
            rm, rv = torch.cat([r.bn.running_mean for r in rs]), torch.cat([r.bn.running_var for r in rs])
        out = ops._BatchNormFn.apply(y, gamma, beta, rm, rv, bn0.num_batches_tracked if bn0.training else None, bn0.momentum, bn0.eps, bn0.training,
                                     int(has_act), None, False)
        if bn0.training:
            with torch.no_grad():
                c = rs[0].conv.out_channels
                torch._foreach_copy_([r.bn.running_mean for r in rs], list(rm.split(c)))
                torch._foreach_copy_([r.bn.running_var for r in rs], list(rv.split(c)))
                torch._foreach_add_([r.bn.num_batches_tracked for r in rs[1:]], 1)
        return list(out.split(rs[0].conv.out_channels, dim=1))

    def get_out_channels(self):
        return [self.tum_out_channels * self.num_tums] * self.num_outputs
