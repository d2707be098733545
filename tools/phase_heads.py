import ctypes, os, sys, runpy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from single_shot_detection_amd import _lib
sys.argv = ['bench_conv.py', '--fwd-only', '--reps', '3']
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_conv.py'), run_name='__main__')
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 256)()
assert raw.ssdk_debug_read_phase(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4).astype(np.int64)
t0 = t[t[:, 0] > 0, 0].min()
for b in range(0, 64, 4):
    print('wg %2d start %7.2f prologue %6.2f kloop %7.2f epilogue %6.2f' % (b, (t[b, 0] - t0) / 100, (t[b, 1] - t[b, 0]) / 100, (t[b, 2] - t[b, 1]) / 100, (t[b, 3] - t[b, 2]) / 100))
