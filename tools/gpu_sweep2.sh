#!/bin/bash
cd $GRAFT_REPO_ROOT
export RT_PERF_PIPES=wf
for v in "RT_WF_INNER_MIN=1" "RT_WF_INNER_MIN=8" "RT_WF_INNER_MIN=24" "RT_WF_INNER_MIN=32" "RT_WF_REFILL=16" "RT_WF_REFILL=24" "RT_WF_REFILL=48" "RT_WF_POOL=16777216" "RT_WF_POOL=4194304"; do
  echo "== $v"; env $v timeout -k 10 120 python tools/gpu_perf.py 2>&1 | grep "f64-wf stats=0" | tail -1 | cut -c1-200
done
