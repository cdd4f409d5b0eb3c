"""Aggregates rocprofv3 --pmc CSV output per kernel: sums each counter over dispatches."""
import csv, glob, sys, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('sa::(anonymous namespace)::', '').replace('void ', ''))
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(name, r['Counter_Name'])] += 1
for name in sorted(agg):
    print(name)
    for c, v in sorted(agg[name].items()):
        print('   %-28s %18.0f  (n=%d)' % (c, v, cnt[(name, c)]))
