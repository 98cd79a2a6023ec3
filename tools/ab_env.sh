#!/bin/bash
# A/B of environment switches of libgitcap in bench.py, interleaved, same box:  tools/ab_env.sh "VAR=1" "VAR=1 OTHER=1" ...
# Each configuration (plus the empty one) runs pipelined and --serial, ROUNDS times, interleaved.
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for cfg in "" "$@"; do
    for mode in "" "--serial"; do
      out=$(env $cfg python bench.py --no-cpu-baseline --plain $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['p50_latency_ms'])")
      echo "round $r  [${cfg:-default}] ${mode:-pipelined}: $out"
    done
  done
done
