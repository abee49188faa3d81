#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { echo -n "== $* : "; env "${@:2}" timeout -k 10 200 python bench.py --workload $1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s, %.0f ms/step' % (d['value'], d['ms_per_step']))"; }
run c2 RT_AUTO_MEGA_NO_MESH=1
run c2 RT_X=1
run c2 RT_WF_SPLIT=0
run c3 RT_X=1
run c4 RT_X=1
