"""Plain-torch backbone stand-ins with torchvision's module indexing (torchvision is not installed in this image and
pretrained weights cannot be fetched).  Out of the hot path (north_star: "backbone stays PyTorch-ROCm"); they exist
so that ``detection.init`` can be driven end to end.  Architectures are the published ones (VGG-16-BN config D;
ResNet-50 bottlenecks; MobileNetV2 inverted residuals); weights are randomly initialised."""
import torch.nn as nn


class _Vgg16Bn(nn.Module):
    CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']

    def __init__(self, pretrained=False, **kwargs):
        super().__init__()
        layers, c = [], 3
        for v in self.CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))  # floor mode: 300 -> 37 -> 18
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)  # 44 modules; out_layers (32, 42) are ReLUs after conv4_3 / conv5_3


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False); self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False); self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        return self.relu(x + idt)


class _ResNet50(nn.Module):
    def __init__(self, pretrained=False, **kwargs):
        super().__init__()

        def stage(cin, width, n, stride):
            return nn.Sequential(*[_Bottleneck(cin if i == 0 else width * 4, width, stride if i == 0 else 1) for i in range(n)])
        # the flattened ``features`` of bf/builders/base_builder.py:10-23: conv1, bn1, relu, maxpool, layer1..4
        self.features = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                      nn.MaxPool2d(3, stride=2, padding=1), stage(64, 64, 3, 1), stage(256, 128, 4, 2),
                                      stage(512, 256, 6, 2), stage(1024, 512, 3, 2))


def _conv_bn_relu6(cin, cout, k, stride, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        hidden = cin * expand
        layers = [] if expand == 1 else [_conv_bn_relu6(cin, hidden, 1, 1)]
        layers += [_conv_bn_relu6(hidden, hidden, 3, stride, groups=hidden), nn.Conv2d(hidden, cout, 1, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)
        self.residual = stride == 1 and cin == cout

    def forward(self, x):
        return x + self.conv(x) if self.residual else self.conv(x)


class _MobileNetV2(nn.Module):
    SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]   # t, c, n, s

    def __init__(self, pretrained=False, **kwargs):
        super().__init__()
        layers, c = [_conv_bn_relu6(3, 32, 3, 2)], 32
        for t, cout, n, s in self.SETTING:
            for i in range(n):
                layers.append(_InvertedResidual(c, cout, s if i == 0 else 1, t))
                c = cout
        layers.append(_conv_bn_relu6(c, 1280, 1, 1))
        self.features = nn.Sequential(*layers)   # 19 modules; samples/ssd_mb2_voc.py taps 13 (96 ch, stride 16) and 18 (1280 ch, stride 32)


_ZOO = {'torchvision_vgg16_bn': _Vgg16Bn, 'torchvision_resnet50': _ResNet50, 'torchvision_mobilenet_v2': _MobileNetV2}


def create_base(name, weight=None, **model_args):
    """bf/builders/base_builder.py:59-86 for the backbones the BASELINE configs name; no remote loaders.

    ``weight``: a state_dict file is loaded like the reference does (:81-84); a path that does not exist raises (the reference skips
    it silently and trains from whatever initialisation the constructor made).  ``pretrained=True`` -- every sample config passes
    it -- meant a torchvision model-zoo download; there is no network and no torchvision here, so the backbone keeps its random
    initialisation and says so once."""
    if name not in _ZOO:
        raise NotImplementedError(f'backbone {name!r} is outside the hot-path scope; available: {sorted(_ZOO)}')
    base = _ZOO[name](**model_args)
    if weight == 'keras':
        raise NotImplementedError("weight='keras' (init_from_keras of the reference's own MobileNets, bf/base/mobilenet*.py) is outside the hot-path scope")
    if weight is not None:
        import os
        import torch
        if not os.path.exists(weight):
            raise FileNotFoundError(f'backbone weight file {weight!r} does not exist')
        base.load_state_dict(torch.load(weight, map_location='cpu'))
    elif model_args.get('pretrained'):
        import logging
        logging.getLogger(__name__).warning(
            "create_base(%r, pretrained=True): no pretrained weights are available offline (torchvision's model zoo is a download); "
            'the backbone is randomly initialised -- pass weight=<state_dict file> to load one', name)
    return base
