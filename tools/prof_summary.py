# Per-step summary of a rocprofv3 --kernel-trace --stats run: prof_summary.py <dir> [steps].
# Launches that belong to handle creation and weight loading (the runtime's fill / copy kernels behind hipMemset / hipMemcpy of the
# workspace and the weights, the one-off fragment-major weight packing) are NOT per-step work: they are listed apart with their totals
# and kept out of the per-step table and its total.  (The data path itself has ONE hipMemsetAsync per call -- the stop counters of the
# greedy loop, a ~2 us fill -- which is counted with them: `steps` of the fillBuffer calls below are that one.)
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
SETUP = ('__amd_rocclr_fillBuffer', '__amd_rocclr_copyBuffer', 'pack_frags_kernel')
def is_setup(r):
    return any(k in r['Name'] for k in SETUP)
setup = [r for r in rows if is_setup(r)]
rows = [r for r in rows if not is_setup(r)]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('%-70s %8s %10s %9s %6s' % ('kernel', 'calls/st', 'ms/step', 'avg us', '%'))
for r in rows[:22]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-70s %8.1f %10.3f %9.2f %5.1f%%' % (n[:70], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('total kernel ms/step', tot / 1e6 / steps)
if setup:
    print('\nset-up launches (handle creation, weight upload / packing: once per process, NOT per step):')
    for r in setup:
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        print('  %-68s %8d calls %10.3f ms total %9.2f avg us' % (n[:68], int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
