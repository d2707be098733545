#!/usr/bin/env python3
"""How busy is the GPU inside the training steps?  From a rocprofv3 rocpd result of bench.py: the dispatches between consecutive
fused-SGD launches are one step; prints per step the span, the sum of kernel durations and the idle share.
    python tools/rocpd_gaps.py <db> [marker substring, default multi_tensor_apply]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else 'multi_tensor_apply'
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f'select d.start, d.end, s.display_name from {disp} d join {sym} s on d.kernel_id = s.id order by d.start').fetchall()
ends = [k for k, r in enumerate(rows) if marker in r[2]]
print('step | launches | span us | busy us | idle %% | largest gaps (us, after kernel)')
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    span = (seg[-1][1] - rows[a][1]) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    gaps = sorted(((seg[k + 1][0] - seg[k][1]) / 1e3, seg[k][2][:40]) for k in range(len(seg) - 1))[-3:]
    print('%3d | %4d | %8.1f | %8.1f | %5.1f | %s' % (ends.index(a), len(seg), span, busy, 100 * (1 - busy / span), '; '.join('%.1f %s' % g for g in reversed(gaps))))
if len(sys.argv) > 3:   # third argument: list the launches of that step
    a, b = ends[int(sys.argv[3])], ends[int(sys.argv[3]) + 1]
    t0 = rows[a][1]
    for s, e, name in rows[a + 1:b + 1]:
        print('%9.1f %8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, name[:90]))
