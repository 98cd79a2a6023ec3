# pipeline shapes: default (encoder stream + 2 decode streams) vs symmetric (3 streams, image pass + token loop of a submission on one stream)
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --plain --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['p50_latency_ms'])" || exit 1; }
ARGS="" run X=1
ARGS="" run GITCAP_PIPE_SYM=1
ARGS="--inflight 4" run GITCAP_PIPE_SYM=1
ARGS="--inflight 2" run GITCAP_PIPE_SYM=1
ARGS="" run X=1
