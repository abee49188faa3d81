#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "RT_WF_POOL=2097152" "RT_WF_POOL=4194304" "RT_WF_POOL=8388608" "RT_WF_POOL=16777216" "RT_WF_POOL=33554432" "RT_WF_REFILL=16" "RT_WF_REFILL=48" "RT_WF_INNER_MIN=8" "RT_WF_INNER_MIN=32"; do
  echo -n "== $v : "; env $v timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s, mesh kernel %.0f ms/step x%d' % (d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches_per_step']))"
done
