#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "$@" timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Msamples/s, %.0f ms/step (mesh %.0f)' % (d['value'], d['ms_per_step'], r['kernel_ms_per_step']))"; }
run RT_WF_SHADE_LDS_PAD=0
run RT_WF_SHADE_LDS_PAD=60000
