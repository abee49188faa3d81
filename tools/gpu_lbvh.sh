#!/bin/bash
cd $GRAFT_REPO_ROOT
RT_BVH_BUILDER=device timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for b in host device; do
  echo "== builder $b"
  RT_BVH_BUILDER=$b RT_COMPILE_DEBUG=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('   %.1f Msamples/s, mesh %.0f ms/step, nodes/ray %.2f tris/ray %.2f' % (d['value'], r['kernel_ms_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray']))
    else: print('  ', l.strip()[:300])"
done
