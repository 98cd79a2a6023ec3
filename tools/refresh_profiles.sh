#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats of bench.py (pipelined and --serial) and the
# three PMC passes (separate runs, kernel-trace only) into gpurun_out/prof_$1/.  Summarise with tools/prof_summary.py,
# tools/pmc_tables.py and copy the results into profiles/.
set -e
tag=${1:-x}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --plain"
rocprofv3 --kernel-trace --stats -d $out/stats_pipelined --output-format csv -- $B > $out/bench_pipelined.json 2> $out/stats_pipelined.err
echo "stats pipelined done"
rocprofv3 --kernel-trace --stats -d $out/stats_serial --output-format csv -- $B --serial > $out/bench_serial.json 2> $out/stats_serial.err
echo "stats serial done"
P="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --plain --serial"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- $P > /dev/null 2> $out/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- $P > /dev/null 2> $out/pmc_write.err
echo "pmc write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $out/pmc_sq --output-format csv -- $P > /dev/null 2> $out/pmc_sq.err
echo "pmc sq done"
# keep only the small CSVs (the merged-back directory is capped at 64 MiB)
find $out -name "*.csv" -size +20M -delete
du -sh $out
