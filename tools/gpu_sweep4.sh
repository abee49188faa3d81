#!/bin/bash
# A/B of the mesh-kernel knobs on the headline config (one step each).
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "$@" timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Msamples/s, mesh %.0f ms/step x%d, nodes/ray %.2f tris/ray %.2f' % (d['value'], r['kernel_ms_per_step'], r['launches_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray']))"; }
run RT_X=0
for l in 6 8 16 24; do run RT_WF_LDS_LEVELS=$l; done
if [ -f gpurun_in_mw5.so ]; then
  for l in 8 12; do run RT_DEVICE_LIB=$GRAFT_REPO_ROOT/gpurun_in_mw5.so RT_WF_LDS_LEVELS=$l; done
fi
run RT_WF_REFILL=16
run RT_WF_REFILL=48
run RT_WF_INNER_MIN=8
run RT_WF_INNER_MIN=24
run RT_WF_POOL=67108864
