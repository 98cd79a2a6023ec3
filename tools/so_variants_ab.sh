#!/bin/bash
# A/B of several BUILDS on one box, one process per run, order rotated: the current gitcap/libgitcap.so ("base") against
# gitcap/libgitcap_<name>.so for every <name>:   tools/so_variants_ab.sh <rounds> "<name> <name> ..." <command...>
# (e.g. builds of one source file with another -mllvm -amdgpu-sched-strategy; the command should print a checksum of its results.)
G=real-time-video-captioning_amd/gitcap
rounds=$1; names="base $2"; shift 2
cp $G/libgitcap.so /tmp/libgitcap_base.so
trap 'cp /tmp/libgitcap_base.so '$G'/libgitcap.so' EXIT
set -- "$@"
arr=($names)
for i in $(seq $rounds); do
  for j in $(seq 0 $((${#arr[@]} - 1))); do
    n=${arr[$(( (i + j) % ${#arr[@]} ))]}
    if [ "$n" = base ]; then cp /tmp/libgitcap_base.so $G/libgitcap.so; else cp $G/libgitcap_$n.so $G/libgitcap.so; fi
    echo "== $n, round $i"; "$@" 2>/dev/null
  done
done
