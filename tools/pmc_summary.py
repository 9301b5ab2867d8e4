"""Average PMC counter values per kernel from a rocprofv3 --pmc CSV directory."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:44] + ' g=' + r.get('Grid_Size', '?')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[k] += 1
names = sorted({c for k in acc for c in acc[k]})
print('%-60s %5s ' % ('kernel', 'n') + ' '.join('%14s' % n[-14:] for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_BUSY_CYCLES', 0)):
    print('%-60s %5d ' % (k, cnt[k]) + ' '.join('%14.0f' % (acc[k][n] / cnt[k]) for n in names))
