#!/bin/bash
# Runs on the GPU box (through gpurun): BASELINE configs[4] (GIT-large, 4 clips x 10 frames, beam 4, 15 steps, e4m3 storage;
# tools/cfg4_run.py) under rocprofv3: kernel-trace stats and the three PMC passes (separate runs) into gpurun_out/prof_cfg4_$1/.
# Summarise: tools/prof_summary.py <dir>/stats 5 ; tools/pmc_tables.py <dir> profiles/rNN_cfg4
set -e
tag=${1:-x}
out=gpurun_out/prof_cfg4_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export PASSES=5
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 tools/cfg4_run.py > /dev/null 2> $out/stats.err
echo "cfg4 stats done"
export PASSES=3
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 tools/cfg4_run.py > /dev/null 2> $out/pmc_fetch.err
echo "cfg4 pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 tools/cfg4_run.py > /dev/null 2> $out/pmc_write.err
echo "cfg4 pmc write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $out/pmc_sq --output-format csv -- python3 tools/cfg4_run.py > /dev/null 2> $out/pmc_sq.err
echo "cfg4 pmc sq done"
find $out -name "*.csv" -size +20M -delete
du -sh $out
