#!/usr/bin/env python3
"""Every call of one kernel, in launch order, from a rocprofv3 rocpd result:  python tools/rocpd_calls.py <db> <kernel name substring>
(why the AVERAGE duration of the head GEMM in a --stats summary sits above the event-timed mean of the timed steps: which calls are slow?)"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select d.start, d.end - d.start from {disp} d join {sym} s on d.kernel_id = s.id where s.display_name like ? order by d.start",
                  ('%' + sys.argv[2] + '%',)).fetchall()
t0 = rows[0][0] if rows else 0
d = [r[1] / 1e3 for r in rows]
for i, (st, du) in enumerate(rows):
    print('%3d  t=%9.3f ms  %8.1f us' % (i, (st - t0) / 1e6, du / 1e3))
if d:
    s = sorted(d)
    print('calls %d  mean %.1f  median %.1f  min %.1f  max %.1f us' % (len(d), sum(d) / len(d), s[len(s) // 2], s[0], s[-1]))
