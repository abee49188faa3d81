#!/bin/bash
# knob sweep for the wavefront pipeline (f64, C4 @ 40 spp)
cd $GRAFT_REPO_ROOT
export RT_PERF_PIPES=wf
for refill in 64 48 32 24 16 8; do
  echo "== REFILL=$refill"; RT_WF_REFILL=$refill timeout -k 10 120 python tools/gpu_perf.py 2>&1 | grep "f64-wf stats=0" | tail -1
done
for pool in 524288 1048576 4194304 8388608; do
  echo "== POOL=$pool"; RT_WF_POOL=$pool timeout -k 10 120 python tools/gpu_perf.py 2>&1 | grep "f64-wf stats=0" | tail -1
done
for chk in 2 4 16 32; do
  echo "== CHECK=$chk"; RT_WF_CHECK=$chk timeout -k 10 120 python tools/gpu_perf.py 2>&1 | grep "f64-wf stats=0" | tail -1
done
