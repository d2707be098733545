#!/usr/bin/env python3
"""Launches of one kernel grouped by grid size (a proxy for the layer): python tools/rocpd_by_grid.py <db> <kernel substring> [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in db.execute(f'pragma table_info({disp})')]
gx = 'grid_size_x' if 'grid_size_x' in cols else [c for c in cols if 'grid' in c][0]
wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else None
rows = db.execute(f'select d.{gx}{", d." + wx if wx else ""}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id '
                  f"where s.display_name like '%{sub}%' group by d.{gx} order by 3 desc").fetchall()
tot = sum(r[-3] for r in rows)
print('grid (threads) | workgroups | calls | total ms | avg us | min us | %')
for r in rows[:top]:
    g = r[0]
    w = r[1] if wx else 256
    n, t, avg, mn = r[-4], r[-3], r[-2], r[-1]
    print('%10d | %6d | %5d | %8.3f | %7.1f | %7.1f | %5.1f' % (g, g // max(1, w), n, t / 1e6, avg / 1e3, mn / 1e3, 100.0 * t / tot))
print('total %.3f ms over %d launches' % (tot / 1e6, sum(r[-4] for r in rows)))
