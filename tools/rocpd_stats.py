#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite result (…_results.db):  python tools/rocpd_stats.py <db> [top] [--md]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f'select s.display_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id group by s.display_name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
print('| kernel | calls | total ms | avg us | min us | % |\n|---|---|---|---|---|---|')
for name, n, t, avg, mn in rows[:top]:
    print('| `%s` | %d | %.3f | %.1f | %.1f | %.2f |' % (name[:100], n, t / 1e6, avg / 1e3, mn / 1e3, 100.0 * t / tot))
print('\ntotal GPU kernel time %.3f ms' % (tot / 1e6))
