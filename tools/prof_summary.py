import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('%-70s %8s %10s %9s %6s' % ('kernel', 'calls/st', 'ms/step', 'avg us', '%'))
for r in rows[:22]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-70s %8.1f %10.3f %9.2f %5.1f%%' % (n[:70], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('total kernel ms/step', tot / 1e6 / steps)
