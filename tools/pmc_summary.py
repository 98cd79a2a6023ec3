# summarise a rocprofv3 --pmc pass per kernel: mean counter value per dispatch
import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:14]:
    print('%-62s' % k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in cs.items()})
