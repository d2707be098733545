"""Conv blocks of the pyramid tail -- mirror of bf/modules/conv.py:4-85 (same submodule names, so state_dict keys
``conv``/``bn``/``activation`` and ``depthwise_*``/``pointwise_*`` line up with the reference's checkpoints)."""
import logging

import torch
import torch.nn as nn

from ... import ops

_warned = set()


def _warn_stock(kind, why):
    """The blocks below run on libssdk for the shapes of the hot path; anything else falls back to the stock PyTorch-ROCm modules
    (neck-side variants, out of scope as kernels).  Said once per reason, not silently."""
    key = (kind, why)
    if key not in _warned:
        _warned.add(key)
        logging.getLogger(__name__).warning('%s: %s -- this block runs on the stock PyTorch-ROCm modules, not on libssdk', kind, why)


def _norm_act(x, bn, act):
    """BatchNorm (+ ReLU) on libssdk for an nn.BatchNorm2d -- per process, or over all ranks when distributed.convert_sync_batchnorm
    marked it (detection.init(distributed=True); reference: apex convert_syncbn_model, detection/init.py:85).  Any other norm module
    (a torch.nn.SyncBatchNorm someone converted by hand, GroupNorm, ...) keeps its own torch kernels."""
    if type(bn) is nn.BatchNorm2d:
        return ops.batch_norm(x, bn, relu=act is not None)
    x = bn(x)
    return act(x) if act is not None else x


class Conv2dBn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1, bias=False, use_bn=True,
                 activation_params={'name': 'ReLU', 'args': {'inplace': True}}, batch_norm_params={}):
        super(Conv2dBn, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              groups=groups, bias=bias)
        if use_bn:
            self.bn = nn.BatchNorm2d(out_channels, **batch_norm_params)
        if activation_params is not None:
            self.activation = getattr(nn, activation_params['name'])(**activation_params['args'])
        # [cout][ky][kx][cin] memory = the rows the implicit GEMM reads (no per-step permute copy); state_dict unchanged
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)

    def _why_not_hip(self):
        c = self.conv
        act = self._modules.get('activation')
        if c.groups != 1:
            return f'groups={c.groups}'
        if c.kernel_size not in ((1, 1), (3, 3)) or c.stride not in ((1, 1), (2, 2)) or c.dilation != (1, 1) or c.padding[0] != c.padding[1] \
                or c.padding_mode != 'zeros':
            return f'kernel_size={c.kernel_size} stride={c.stride} dilation={c.dilation} padding={c.padding} ({c.padding_mode})'
        if c.in_channels % 4 or c.out_channels % 4:
            return f'channels {c.in_channels}->{c.out_channels} not multiples of 4'
        if act is not None and not isinstance(act, nn.ReLU):
            return f'activation {type(act).__name__}'
        return None

    def _hip_ok(self):
        return self._why_not_hip() is None

    def forward(self, x):  # conv.py:30-36: conv -> BN -> activation
        if self._hip_ok():   # libssdk: implicit-GEMM conv (csrc/conv.hip) + BatchNorm/ReLU kernels (csrc/norm.hip)
            has_bn, has_act = 'bn' in self._modules, 'activation' in self._modules
            c = self.conv
            if has_bn and type(self.bn) is nn.BatchNorm2d:   # (statistics in the convolution's epilogue where they can be: ops.conv2d_batch_norm)
                return ops.conv2d_batch_norm(x, c.weight, c.bias, c.stride[0], c.padding[0], self.bn, conv_relu=False, bn_relu=has_act)
            x = ops.conv2d(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0], relu=has_act and not has_bn)
            if has_bn:
                x = _norm_act(x, self.bn, self._modules.get('activation'))
            return x
        # grouped / exotic variants (neck-side, out of the hot-path scope): stock PyTorch-ROCm modules
        _warn_stock('Conv2dBn', self._why_not_hip())
        x = self.conv(x)
        if 'bn' in self._modules:
            x = self.bn(x)
        if 'activation' in self._modules:
            x = self.activation(x)
        return x


class DepthwiseConv2dBn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, use_bn=True,
                 activation_params={'name': 'ReLU', 'args': {'inplace': True}}, batch_norm_params={}):
        super(DepthwiseConv2dBn, self).__init__()
        self.depthwise_conv = nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                        groups=in_channels, bias=bias)
        if use_bn:
            self.depthwise_bn = nn.BatchNorm2d(in_channels, **batch_norm_params)
        if activation_params is not None:
            self.depthwise_activation = getattr(nn, activation_params['name'])(**activation_params['args'])
        self.pointwise_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)
        self.pointwise_conv.weight.data = self.pointwise_conv.weight.data.contiguous(memory_format=torch.channels_last)
        if use_bn:
            self.pointwise_bn = nn.BatchNorm2d(out_channels, **batch_norm_params)
        if activation_params is not None:
            self.pointwise_activation = getattr(nn, activation_params['name'])(**activation_params['args'])

    def _hip_ok(self):
        d, pw = self.depthwise_conv, self.pointwise_conv
        acts = [self._modules.get('depthwise_activation'), self._modules.get('pointwise_activation')]
        return (d.kernel_size[0] == d.kernel_size[1] and d.kernel_size[0] <= 5 and d.stride[0] == d.stride[1]
                and d.padding[0] == d.padding[1] and d.dilation == (1, 1) and d.padding_mode == 'zeros' and d.in_channels % 4 == 0
                and pw.out_channels % 4 == 0 and all(a is None or isinstance(a, nn.ReLU) for a in acts))

    def forward(self, x):  # conv.py:72-85: depthwise conv -> BN -> act -> pointwise conv -> BN -> act
        if self._hip_ok():   # libssdk: depthwise stencil (csrc/norm.hip) + 1x1 implicit GEMM (csrc/conv.hip) + BatchNorm/ReLU kernels
            has_bn = 'depthwise_bn' in self._modules
            d, pw = self.depthwise_conv, self.pointwise_conv
            x = ops.depthwise_conv2d(x, d.weight, d.bias, stride=d.stride[0], padding=d.padding[0])
            if has_bn:
                x = _norm_act(x, self.depthwise_bn, self._modules.get('depthwise_activation'))
            elif 'depthwise_activation' in self._modules:
                x = torch.relu(x)
            x = ops.conv2d(x, pw.weight, pw.bias, relu='pointwise_activation' in self._modules and not has_bn)
            if has_bn:
                x = _norm_act(x, self.pointwise_bn, self._modules.get('pointwise_activation'))
            return x
        _warn_stock('DepthwiseConv2dBn', 'kernel / stride / padding / channel count / activation outside what the depthwise kernels take')
        for name in ('depthwise_conv', 'depthwise_bn', 'depthwise_activation', 'pointwise_conv', 'pointwise_bn',
                     'pointwise_activation'):
            if name in self._modules:
                x = self._modules[name](x)
        return x
