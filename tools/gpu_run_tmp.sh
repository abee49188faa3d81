#!/bin/bash
source tools/gpu_steps.sh
mkdir -p gpurun_out/r02_workloads
for w in c1 c2 c3 c4; do for p in f64 f32; do
  step r02_workloads/bench_${w}_${p} 300 python bench.py --workload $w --precision $p --steps 3 --warmup 1 --no-cpu-baseline
  python3 -c "
import json
l=[x for x in open('gpurun_out/r02_workloads/bench_${w}_${p}.log') if x.startswith('{')]
if l:
    d=json.loads(l[0]); r=d['roofline'] or {}
    print('$w $p', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms', r.get('kernel','')[:12], r.get('frac'), r.get('all_kernels_ms_per_step'))
else: print('$w $p FAILED')
"
done; done
