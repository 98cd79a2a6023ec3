# gemm256 beside hipBLASLt's kernel, shape by shape, from the passes of tools/gemm_vs_library_pmc.sh:
#   python tools/pmc_gemm_vs_library.py gpurun_out/prof_lib_r6 profiles/r06_gemm_vs_library_pmc.md
# tools/gemm_vs_library.py runs, per shape, 33 launches of gemm256, then 33 of F.linear (library, bias epilogue), then 33 of mm
# (library, no epilogue): consecutive dispatches of one kernel form a segment, segments are labelled in that order.
# FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM);
# GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
SHAPES = [(768, 768, 0), (768, 768, 3), (2304, 768, 0), (3072, 768, 0), (3072, 768, 1), (3072, 768, 2), (768, 3072, 3), (768, 3072, 0)]
EPI = {0: 'bias -> bf16', 1: 'bias + QuickGELU -> bf16', 2: 'bias + erf GELU -> bf16', 3: 'bias + fp32 residual -> fp32'}
def load(sub):
    f = glob.glob('%s/%s/**/*counter_collection.csv' % (src, sub), recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        if not (n.startswith('gemm256_kernel') or 'Cijk' in n): continue
        d = per.setdefault(int(r['Dispatch_Id']), {'name': n, 'grid': int(r['Grid_Size']), 'wg': r['Workgroup_Size'], 'lds': r['LDS_Block_Size'],
                                                   'vgpr': r['VGPR_Count'], 'us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'c': {}})
        d['c'][r['Counter_Name']] = float(r['Counter_Value'])
    segs = []
    for did in sorted(per):
        d = per[did]
        if not segs or segs[-1][0]['name'] != d['name'] or segs[-1][0]['grid'] != d['grid'] or len(segs[-1]) >= 33: segs.append([])
        segs[-1].append(d)
    return segs
fe, wr, sq = load('pmc_fetch'), load('pmc_write'), load('pmc_sq')
mean = lambda v: sum(v) / len(v) if v else 0.0
assert len(fe) == len(wr) == len(sq) == 3 * len(SHAPES), (len(fe), len(wr), len(sq))
lines = ['# gemm256 beside hipBLASLt on the same operands: kernel identity and counters (round 6)\n',
         'M = 18 944, bf16 operands.  rocprofv3 --pmc, three separate passes of tools/gemm_vs_library.py (tools/gemm_vs_library_pmc.sh); times under the',
         'counters are a few % above the unprofiled ones (gpurun_out/prof_lib_r6/times.txt has those).  The library kernel for every shape:',
         '`%s` -- a hand-written ("Custom") 256 x 256 x 64 macro-tile kernel, 256 threads = one wave per SIMD (128 x 128 accumulators per wave),' % next(s[0]['name'] for s in fe if 'Cijk' in s[0]['name']),
         '%s B of LDS, stream-K (SK3): a PERSISTENT grid of %d workgroups whatever the shape (222 of the 256 CUs), each walking an equal share of the' % (next(s[0]['lds'] for s in fe if 'Cijk' in s[0]['name']), next(s[0]['grid'] for s in fe if 'Cijk' in s[0]['name']) // 256),
         'K-iterations of all tiles, so the next tile\'s operands are in flight under the previous tile\'s epilogue.\n',
         '| shape (N, K) | kernel | epilogue | avg us | read MB (2 x FETCH) | write MB | MFMA busy | LDS conflict / active |', '|---|---|---|---|---|---|---|---|']
for i, (N, K, epi) in enumerate(SHAPES):
    for j, who in enumerate(('gemm256', 'hipBLASLt (F.linear)', 'hipBLASLt (mm)')):
        a, b, c = fe[3 * i + j], wr[3 * i + j], sq[3 * i + j]
        gui = mean([d['c']['GRBM_GUI_ACTIVE'] for d in c]) / 8.0
        util = mean([d['c']['SQ_VALU_MFMA_BUSY_CYCLES'] for d in c]) / (gui * 1024.0)
        lds = mean([d['c']['SQ_LDS_BANK_CONFLICT'] for d in c]) / max(1.0, mean([d['c']['SQ_LDS_IDX_ACTIVE'] for d in c]))
        ep = EPI[epi] if j == 0 else ('bias -> bf16' if j == 1 else 'none (bf16 out)')
        lines.append('| %d, %d | %s | %s | %.1f | %.1f | %.1f | %.1f %% | %.1f %% |' % (N, K, who, ep, mean([d['us'] for d in a]),
                     2.0 * mean([d['c']['FETCH_SIZE'] for d in a]) * 1024 / 1e6, mean([d['c']['WRITE_SIZE'] for d in b]) * 1024 / 1e6, 100 * util, 100 * lds))
open(dst, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
