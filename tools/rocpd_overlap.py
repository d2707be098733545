#!/usr/bin/env python3
"""Two-stream timeline of one training step from a rocprofv3 rocpd result of bench.py: every dispatch between two fused-SGD launches
with its start (us from the step's begin), duration and HIP stream / HSA queue, plus how much of the step's span had kernels of two
streams in flight at once.
    python tools/rocpd_overlap.py <db> [step index, default 5] [marker substring, default multi_tensor_apply]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 5
marker = sys.argv[3] if len(sys.argv) > 3 else 'multi_tensor_apply'
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in db.execute(f'pragma table_info({disp})')]
lane = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
sel = f'd.{lane}' if lane else '0'
rows = db.execute(f'select d.start, d.end, s.display_name, {sel}, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} s on d.kernel_id = s.id order by d.start').fetchall()
ends = [k for k, r in enumerate(rows) if marker in r[2]]
a, b = ends[step], ends[step + 1]
seg = rows[a + 1:b + 1]
t0 = rows[a][1]
lanes = sorted({r[3] for r in seg})
print(f'# columns of {disp}: {cols}')
print(f'# step {step}: {len(seg)} dispatches, span {(seg[-1][1] - t0) / 1e3:.1f} us, lanes ({lane}) {lanes}')
# time with >= 2 lanes busy
evs = []
for s, e, _, ln, _, _ in seg:
    evs.append((s, 1, ln))
    evs.append((e, -1, ln))
evs.sort()
busy = {ln: 0 for ln in lanes}
last, both, any_ = evs[0][0], 0, 0
for t, d, ln in evs:
    n = sum(1 for v in busy.values() if v > 0)
    if n >= 2:
        both += t - last
    if n >= 1:
        any_ += t - last
    last = t
    busy[ln] += d
print(f'# some kernel in flight {any_ / 1e3:.1f} us, kernels of two lanes in flight {both / 1e3:.1f} us')
for s, e, name, ln, gx, wx in seg:
    print('%9.1f %8.1f  lane %-3s grid %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, lanes.index(ln), gx // max(wx, 1), name[:80]))
