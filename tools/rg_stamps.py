#!/usr/bin/env python3
"""Experiment: where the row GEMM of the ordered heads backward spends its time (per workgroup: prologue / MFMA section / wait + barrier /
store + DMA issue), from 100 MHz wall-clock stamps.  Needs a library built with -DSSDK_RG_STAMPS and selected with SSDK_LIB:
  cd single_shot_detection_amd/csrc && mkdir -p build/stamps && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSSDK_RG_STAMPS -c conv.hip -o build/stamps/conv.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/build/libssdk_stamps.so build/stamps/conv.o $(ls build/*.o | grep -v conv.o)
  SSDK_LIB=tools/build/libssdk_stamps.so python tools/rg_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from single_shot_detection_amd import _lib  # noqa: E402
raw = ctypes.CDLL(_lib.LIB_PATH)
hp = bench.HotPath('ssd_300_vgg16_voc', 32, torch.device('cuda:0'))
for _ in range(3):
    hp.train_step()
torch.cuda.synchronize()
assert raw.ssdk_debug_zero_rg_stamps() == 0
hp.train_step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 8))()
assert raw.ssdk_debug_read_rg_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
live = t[:, 5] > 0
print('workgroups with items', int(live.sum()), 'items', int(t[:, 5].sum()), 'steps', int(t[:, 6].sum()))
for name, col in (('prologue', 0), ('MFMA section', 1), ('wait + barrier', 2), ('stores + DMA issue', 3), ('item total', 4)):
    v = t[live, col] / 100.0
    print(f'{name:20s} us per workgroup: mean {v.mean():7.2f}  min {v.min():7.2f}  max {v.max():7.2f}   per item {t[live, col].sum() / t[live, 5].sum() / 100:.2f}  per step {t[live, col].sum() / t[live, 6].sum() / 100:.3f}')
