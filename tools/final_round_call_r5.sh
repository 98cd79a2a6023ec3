# Round 5: one gpurun call that re-validates HEAD and refreshes the round's profile set (profiles/r05_*):
# the whole -m gpu suite, rocprofv3 kernel-trace stats (pipelined / serial / serial on 256-row tiles) + the three PMC passes of
# bench.py, the PMC tables (so that bench.py sees a fresh `traffic`), the driver-style bench.py, the configs[4] traces
# (synchronous and PIPELINED: tools/cfg4_pipeline.py) and the single-clip trace.
set -e
T=r5f
python -m pytest tests -x -q -m gpu > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
bash tools/refresh_profiles.sh $T > gpurun_out/${T}_refresh.log 2>&1 || { tail -20 gpurun_out/${T}_refresh.log; exit 1; }
python tools/pmc_tables.py gpurun_out/prof_$T profiles/r05 > gpurun_out/${T}_pmc_tables.log 2>&1
cp profiles/r05_pmc_gemm.json gpurun_out/${T}_pmc_gemm.json; cp profiles/r05_pmc_summary.md gpurun_out/${T}_pmc_summary.md
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 600 gpurun_out/${T}_bench.json | head -c 300; echo
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_b1 --output-format csv -- python3 tools/b1_run.py > /dev/null 2> gpurun_out/${T}_b1.err
export COMPUTE=fp8_ffn STORAGE=fp8_e4m3 PASSES=5
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_cfg4_sync --output-format csv -- python3 tools/cfg4_run.py > /dev/null 2> gpurun_out/${T}_cfg4_sync.err
export PIPE=1 PASSES=12
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_cfg4_pipelined --output-format csv -- python3 tools/cfg4_pipeline.py > /dev/null 2> gpurun_out/${T}_cfg4_pipe.err
unset PIPE PASSES COMPUTE STORAGE
python tools/cfg4_pipeline.py > gpurun_out/${T}_cfg4_pipeline.txt 2>&1
find gpurun_out/prof_$T -name "*.csv" -size +20M -delete
du -sh gpurun_out
