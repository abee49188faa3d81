#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $* : "; env "$@" RT_WF_DEBUG=1 timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('   %.1f Msamples/s, mesh %.0f ms/step x%d, nodes/ray %.2f tris/ray %.2f' % (d['value'], r['kernel_ms_per_step'], r['launches_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray']))
    else: print('  ', l.strip()[:300])"; }
run RT_X=0
run RT_WF_LDS_LEVELS=8
run RT_WF_INNER_MIN=8
run RT_WF_INNER_MIN=24
run RT_WF_INNER_MIN=32
run RT_BVH_MAX_LEAF=6
run RT_BVH_MAX_LEAF=8
if [ -f gpurun_in_mw4.so ]; then
  run RT_DEVICE_LIB=$GRAFT_REPO_ROOT/gpurun_in_mw4.so RT_WF_LDS_LEVELS=12
fi
