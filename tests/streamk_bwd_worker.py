"""Child process of tests/test_conv_bn_gpu.py::test_streamk_data_gradient_equals_the_whole_tile_launch (SSDK_CONV_STREAMK_BWD=1 is read once
per process).  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import _lib, ops  # noqa: E402


def main():
    rng = np.random.default_rng(21)
    B, Cc, H = 8, 256, 96
    x = torch.from_numpy(rng.standard_normal((B, Cc, H, H), dtype=np.float32)).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.from_numpy((rng.standard_normal((Cc, Cc, 3, 3)) * 0.02).astype(np.float32)).cuda().contiguous(memory_format=torch.channels_last)
    dy = torch.from_numpy(rng.standard_normal((B, Cc, H, H), dtype=np.float32)).cuda().contiguous(memory_format=torch.channels_last)

    def run():
        xs = x.clone().requires_grad_(True)
        ws = w.clone().requires_grad_(True)
        (y,) = ops.conv2d([xs], ws, None, stride=1, padding=1)
        y.backward(dy)
        return xs.grad.clone(), ws.grad.clone()

    dx_sk, dw_sk = run()
    os.environ['SSDK_CONV_NO_STREAMK'] = '1'     # (read per call: the whole-tile launch)
    dx_pl, dw_pl = run()
    del os.environ['SSDK_CONV_NO_STREAMK']
    ref = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=1, padding=1)
    scale = float(dx_pl.abs().max())
    print(json.dumps({'timeouts': _lib.streamk_timeouts(), 'differs': not torch.equal(dx_sk, dx_pl),
                      'max_vs_plain': float((dx_sk - dx_pl).abs().max()) / scale,
                      'dw_max_vs_plain': float((dw_sk - dw_pl).abs().max()) / float(dw_pl.abs().max()),
                      'max_vs_torch': float((dx_sk - ref).abs().max()) / scale}))


if __name__ == '__main__':
    main()
