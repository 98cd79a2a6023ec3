set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r4f_tests.log 2>&1 || { tail -30 gpurun_out/r4f_tests.log; exit 1; }
tail -2 gpurun_out/r4f_tests.log
bash tools/refresh_profiles.sh r4f > gpurun_out/r4f_refresh.log 2>&1 || { tail -20 gpurun_out/r4f_refresh.log; exit 1; }
python tools/pmc_tables.py gpurun_out/prof_r4f profiles/r04 > gpurun_out/r4f_pmc_tables.log 2>&1
cp profiles/r04_pmc_gemm.json gpurun_out/r4f_pmc_gemm.json; cp profiles/r04_pmc_summary.md gpurun_out/r4f_pmc_summary.md
python bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err
tail -c 600 gpurun_out/r4f_bench.json | head -c 300; echo
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4f/stats_b1 --output-format csv -- python3 tools/b1_run.py > /dev/null 2> gpurun_out/r4f_b1.err
find gpurun_out/prof_r4f -name "*.csv" -size +20M -delete
du -sh gpurun_out
