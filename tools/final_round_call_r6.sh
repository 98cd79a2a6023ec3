# Round 6: re-validates HEAD and refreshes the round's profile set (profiles/r06_*), as TWO gpurun calls (a call is capped at 20 minutes):
#   bash tools/final_round_call_r6.sh A   the whole -m gpu suite, the driver-style bench.py (with the CPU baseline), the stress-divergence table
#   bash tools/final_round_call_r6.sh B   rocprofv3 kernel-trace stats (pipelined / serial) + the three PMC passes of bench.py, the PMC tables,
#                                         the configs[4] traces (synchronous and pipelined), the single-clip and configs[1] traces
set -e
T=r6f
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$1" = "A" ]; then
  python -m pytest tests -x -q -m gpu > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
  tail -2 gpurun_out/${T}_tests.log
  python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  tail -c 600 gpurun_out/${T}_bench.json | head -c 300; echo
  python tools/stress_divergence.py gpurun_out/${T}_stress_divergence.md > gpurun_out/${T}_div.log 2>&1
  tail -2 gpurun_out/${T}_div.log
else
  bash tools/refresh_profiles.sh $T > gpurun_out/${T}_refresh.log 2>&1 || { tail -20 gpurun_out/${T}_refresh.log; exit 1; }
  python tools/pmc_tables.py gpurun_out/prof_$T profiles/r06 > gpurun_out/${T}_pmc_tables.log 2>&1
  cp profiles/r06_pmc_gemm.json gpurun_out/${T}_pmc_gemm.json; cp profiles/r06_pmc_summary.md gpurun_out/${T}_pmc_summary.md
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_b1 --output-format csv -- python3 tools/b1_run.py > /dev/null 2> gpurun_out/${T}_b1.err
  export B=32 F=1
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_cfg1 --output-format csv -- python3 tools/b1_run.py > /dev/null 2> gpurun_out/${T}_cfg1.err
  unset B F
  export COMPUTE=fp8_ffn STORAGE=fp8_e4m3 PASSES=5
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_cfg4_sync --output-format csv -- python3 tools/cfg4_run.py > /dev/null 2> gpurun_out/${T}_cfg4_sync.err
  export PIPE=1 PASSES=12
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T/stats_cfg4_pipelined --output-format csv -- python3 tools/cfg4_pipeline.py > /dev/null 2> gpurun_out/${T}_cfg4_pipe.err
  unset PIPE PASSES COMPUTE STORAGE
  find gpurun_out/prof_$T -name "*.csv" -size +20M -delete
  du -sh gpurun_out/prof_$T
fi
