#!/bin/bash
# Runs on the GPU box: tools/gemm_vs_library.py (gemm256 and hipBLASLt on the same operands) under rocprofv3 -- a kernel trace with
# stats (kernel names = the library's Tensile / custom-kernel configuration) and the three PMC passes (separate runs, kernel-trace
# only), into gpurun_out/prof_lib_$1/.  Summarise with tools/pmc_gemm_vs_library.py.
set -e
tag=${1:-x}
out=gpurun_out/prof_lib_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
P="python3 tools/gemm_vs_library.py"
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- $P > $out/times.txt 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- $P > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- $P > /dev/null 2> $out/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $out/pmc_sq --output-format csv -- $P > /dev/null 2> $out/pmc_sq.err
find $out -name "*.csv" -size +20M -delete
du -sh $out
