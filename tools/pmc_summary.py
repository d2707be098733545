"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if sub and sub not in r['Kernel_Name']: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print('   %-32s %16.0f  per launch over %d launches' % (n, v / len(cnt[(k, n)]), len(cnt[(k, n)])))
