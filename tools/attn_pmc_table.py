# Per-launch SQ counter table of attn_full_kernel from the passes of tools/attn_pmc.sh:
#   python tools/attn_pmc_table.py gpurun_out/prof_attn_x > profiles/r04_attn_full_pmc.md
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
# summed over SIMDs (MI355X_MICROARCH.md, cycle constants).  The two shapes are told apart by their grid size.
import csv, glob, sys, collections
src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for sub in ('pmc_a', 'pmc_b', 'pmc_c'):
    fs = glob.glob('%s/%s/**/*counter_collection.csv' % (src, sub), recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        if 'attn_full_kernel' not in r['Kernel_Name']:
            continue
        shape = 'S=197 (96 frames x 12 heads, grid %s)' % r['Grid_Size'] if int(r['Grid_Size']) // 256 == 2 * 12 * 96 else 'S=1182 (16 clips x 12 heads, grid %s)' % r['Grid_Size']
        acc[shape][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[shape][(sub, r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
mean = lambda v: sum(v) / len(v) if v else float('nan')
print('# attn_full_kernel: SQ counters per launch (rocprofv3 --pmc, three passes of tools/attn_pmc.sh; tools/attn_bench.py)\n')
for shape in sorted(acc):
    c = {k: mean(v) for k, v in acc[shape].items()}
    us = mean(list(dur[shape].values()))
    print('## %s, %.1f us per launch under the profiler\n' % (shape, us))
    print('| counter | per launch | share of SQ_WAVE_CYCLES |\n|---|---|---|')
    wc = c.get('SQ_WAVE_CYCLES', float('nan'))
    for k in sorted(c):
        share = '%.1f %%' % (100 * c[k] / wc) if k.startswith(('SQ_WAIT', 'SQ_ACTIVE_INST', 'SQ_INST_CYCLES')) and wc == wc else ''
        print('| %s | %.4g | %s |' % (k, c[k], share))
    if 'SQ_INSTS_VALU' in c and 'SQ_INSTS_MFMA' in c:
        print('\nVALU instructions per MFMA instruction: %.1f' % (c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        print('MFMA pipe busy: %.1f %% of SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))' % (100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)))
    print()
