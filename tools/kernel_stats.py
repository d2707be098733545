#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary (…kernel_stats.csv) -> markdown table:  python tools/kernel_stats.py <csv> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
    print('| `%s` | %s | %.3f | %.1f | %.2f |' % (r['Name'][:96], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('\ntotal GPU kernel time %.3f ms' % (tot / 1e6))
