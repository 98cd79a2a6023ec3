#!/bin/bash
# A/B of two BUILDS on one box: gitcap/libgitcap.so (new) against gitcap/libgitcap_prev.so (built from the previous sources, e.g.
# `git stash; make; cp libgitcap.so libgitcap_prev.so; git stash pop; make`), interleaved:  tools/ab_prev_so.sh <rounds> <command...>
G=real-time-video-captioning_amd/gitcap
rounds=$1; shift
cp $G/libgitcap.so /tmp/libgitcap_new.so
for i in $(seq $rounds); do
  cp /tmp/libgitcap_new.so $G/libgitcap.so; echo "== new build, round $i"; "$@" 2>/dev/null
  cp $G/libgitcap_prev.so $G/libgitcap.so; echo "== previous build, round $i"; "$@" 2>/dev/null
done
cp /tmp/libgitcap_new.so $G/libgitcap.so
