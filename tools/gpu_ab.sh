#!/bin/bash
cd $GRAFT_REPO_ROOT
export RT_PERF_PIPES=wf
for lib in "$@"; do
  echo "== $lib"; RT_DEVICE_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 120 python tools/gpu_perf.py 2>&1 | grep -E "f64-wf stats=0|rror" | tail -1
done
