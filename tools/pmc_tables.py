# Per-kernel PMC table + the GEMM roofline JSON from the three rocprofv3 --pmc passes of tools/refresh_profiles.sh:
#   python tools/pmc_tables.py gpurun_out/prof_e profiles/r01_e
# FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM);
# GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import library_source_sha   # fingerprint of the kernel sources these counters were measured on
src, dst = sys.argv[1], sys.argv[2]
def load(sub):
    f = glob.glob('%s/%s/**/*counter_collection.csv' % (src, sub), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return acc, dur
fe, dur = load('pmc_fetch'); wr, _ = load('pmc_write'); sq, _ = load('pmc_sq')
mean = lambda v: sum(v) / len(v) if v else 0.0
rows = []
for k in fe:
    rd = 2.0 * mean(fe[k]['FETCH_SIZE']) * 1024 / 1e6; w = mean(wr[k]['WRITE_SIZE']) * 1024 / 1e6; us = mean(dur[k])
    gui = mean(sq[k]['GRBM_GUI_ACTIVE']) / 8.0
    util = mean(sq[k]['SQ_VALU_MFMA_BUSY_CYCLES']) / (gui * 1024.0) if gui else 0.0
    lds = mean(sq[k]['SQ_LDS_BANK_CONFLICT']) / mean(sq[k]['SQ_LDS_IDX_ACTIVE']) if mean(sq[k]['SQ_LDS_IDX_ACTIVE']) else 0.0
    rows.append((len(dur[k]) * us, k, len(dur[k]), rd, w, us, (rd + w) / us if us else 0, util, lds))
rows.sort(reverse=True)
with open(dst + '_pmc_summary.md', 'w') as f:
    f.write('# PMC summary (rocprofv3 --pmc, separate passes; bench.py --serial, B=16 F=6 T=20)\n\n'
            'FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM);\n'
            'GRBM_GUI_ACTIVE is the sum over the 8 XCDs. HBM-side traffic includes Infinity-Cache hits.\n'
            'MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs).\n\n'
            '| kernel | launches | read MB/launch (2x FETCH) | write MB/launch | avg us (profiled) | traffic TB/s | MFMA util | LDS conflict / active |\n|---|---|---|---|---|---|---|---|\n')
    for _, k, n, rd, w, us, tb, util, lds in rows[:18]:
        f.write('| %s | %d | %.1f | %.1f | %.1f | %.2f | %.1f %% | %.1f %% |\n' % (k[:48], n, rd, w, us, tb, 100 * util, 100 * lds))
g = [k for k in fe if k.startswith('gemm256_kernel')]
cat = lambda d, c: [v for k in g for v in d[k][c]]
fetch, write = mean(cat(fe, 'FETCH_SIZE')), mean(cat(wr, 'WRITE_SIZE'))
busy, gui = mean(cat(sq, 'SQ_VALU_MFMA_BUSY_CYCLES')), mean(cat(sq, 'GRBM_GUI_ACTIVE'))
js = {"kernel": "gemm256_kernel (all epilogues), bench.py --serial (one batch at a time; the kernels are those of the pipelined path), 4 steps", "source_sha": library_source_sha(), "launches_sampled": len(cat(fe, 'FETCH_SIZE')),
      "FETCH_SIZE_KB_avg_raw": round(fetch, 1), "WRITE_SIZE_KB_avg": round(write, 1), "gfx950_fetch_correction": 2.0,
      "traffic_bytes_per_launch": int((2.0 * fetch + write) * 1024), "mfma_busy_cycles_avg": int(busy),
      "grbm_gui_active_sum8xcd_avg": int(gui),
      "lds_bank_conflict_over_idx_active": round(mean(cat(sq, 'SQ_LDS_BANK_CONFLICT')) / mean(cat(sq, 'SQ_LDS_IDX_ACTIVE')), 4),
      "mfma_util": round(busy / (gui / 8.0 * 1024.0), 4),
      "commands": ["tools/refresh_profiles.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES, one pass each, -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --serial)"]}
json.dump(js, open(dst + '_pmc_gemm.json', 'w'), indent=1)
print(open(dst + '_pmc_summary.md').read()); print(json.dumps(js)[:400])
