#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc csv passes:  python tools/pmc_table.py <dir>... [--filter substr]"""
import collections, csv, glob, sys
dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
flt = None
if '--filter' in sys.argv:
    flt = sys.argv[sys.argv.index('--filter') + 1]
    dirs = [d for d in dirs if d != flt]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            if flt and flt not in k:
                continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for name, vals in sorted(cs.items()):
        vals = vals[1:] if len(vals) > 2 else vals
        print('   %-28s %16.0f  (n=%d)' % (name, sum(vals) / len(vals), len(vals)))
