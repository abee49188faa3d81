#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "$@" timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Msamples/s, mesh %.0f ms/step x%d' % (d['value'], r['kernel_ms_per_step'], r['launches_per_step']))"; }
for b in 1 2 3 4; do run RT_WF_MESH_BLOCKS=$b; done
