# A/B of one environment switch of libgitcap: bash tools/ab_env.sh VAR  (runs bench.py pipelined + --serial with VAR unset, then VAR=1)
V=$1
for mode in off on; do
  if [ $mode = on ]; then export $V=1; else unset $V; fi
  echo "== $V $mode"
  timeout -k 10 200 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
  timeout -k 10 200 python bench.py --serial 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'])" || exit 1
done
