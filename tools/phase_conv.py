#!/usr/bin/env python3
"""Experiment: where a small convolution's workgroups spend their time (prologue / K loop / epilogue), from wall-clock stamps.
Needs a library built with -DSSDK_CONV_PHASE and selected with SSDK_LIB:
  cd single_shot_detection_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSSDK_CONV_PHASE -c conv.hip -o build/phase/conv.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/build/libssdk_phase.so build/phase/conv.o <the other build/*.o>
  SSDK_LIB=tools/build/libssdk_phase.so python tools/phase_conv.py [bwd]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import ops, _lib  # noqa: E402
dev = torch.device('cuda')
raw = ctypes.CDLL(_lib.LIB_PATH)
bwd = len(sys.argv) > 1 and sys.argv[1] == 'bwd'
# the six small layers of the SSD-300 pyramid tail (samples/ssd_300_vgg16_voc.py: all blocks 's' = 1 x 1, then 3 x 3 / 2 pad 1)
for cin, cout, k, s, p, h in [(512, 128, 1, 1, 0, 10), (128, 256, 3, 2, 1, 10), (256, 128, 1, 1, 0, 5), (128, 256, 3, 2, 1, 5),
                              (256, 128, 1, 1, 0, 3), (128, 256, 3, 2, 1, 3)]:
    x = torch.randn(32, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(bwd)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.01).contiguous(memory_format=torch.channels_last)   # (data gradient only)
    for _ in range(3):
        if bwd:
            y = ops.conv2d(x, w, None, s, p)
            y.backward(torch.ones_like(y))   # the LAST conv kernel of the backward that stamps is the data gradient or the weight gradient's
        else:
            with torch.no_grad(): ops.conv2d(x, w, None, s, p)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    assert raw.ssdk_debug_read_phase(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4).astype(np.int64)
    live = t[:, 0] > t[:, 0].max() - 5000   # (stamps of this launch: the buffer keeps those of earlier, larger launches; 100 MHz clock)
    t0 = t[live, 0].min()
    print(f'{cin}->{cout} k{k} s{s} {h}x{h} {"backward-data" if bwd else "forward"}: {int(live.sum())}{"+" if live.all() else ""} workgroups stamped, '
          f'first start to last end {(t[live, 3].max() - t0) / 100:.2f} us; us relative to the first workgroup start')
    for b in np.nonzero(live)[0][:8]:
        print('  wg %2d start %6.2f prologue %6.2f kloop %6.2f epilogue %6.2f' % (b, (t[b, 0] - t0) / 100, (t[b, 1] - t[b, 0]) / 100, (t[b, 2] - t[b, 1]) / 100, (t[b, 3] - t[b, 2]) / 100))
