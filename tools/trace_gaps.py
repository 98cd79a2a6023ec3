# per-queue analysis of a rocprofv3 kernel trace: for each kernel type, its duration and the gap
# between the end of the previous kernel on the same queue and its own start (dispatch wait)
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
stats = collections.defaultdict(lambda: [0, 0.0, 0.0])
for q, ks in byq.items():
    ks.sort()
    print('queue', q, 'kernels', len(ks), 'span ms', (ks[-1][1] - ks[0][0]) / 1e6)
    for (s0, e0, _), (s1, e1, n1) in zip(ks, ks[1:]):
        n = n1.replace('(anonymous namespace)::', '').replace('void ', '')[:40]
        st = stats[n]; st[0] += 1; st[1] += (e1 - s1) / 1e3; st[2] += max(0, s1 - e0) / 1e3
print('%-42s %7s %10s %10s' % ('kernel', 'n', 'avg dur us', 'avg gap us'))
for n, (c, d, g) in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print('%-42s %7d %10.2f %10.2f' % (n, c, d / c, g / c))
