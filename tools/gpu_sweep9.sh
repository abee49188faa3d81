#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "$@" timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Msamples/s, mesh %.0f ms/step, nodes/ray %.2f tris/ray %.2f' % (d['value'], r['kernel_ms_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray']))"; }
for b in 8 16 32 64; do run RT_BVH_BINS=$b; done
