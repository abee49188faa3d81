#!/bin/bash
# Final verification + the workload table of DESIGN 6.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in c2 c3 c4; do for p in f64 f32; do
  timeout -k 10 200 python bench.py --workload $w --precision $p --no-cpu-baseline 2>/dev/null > gpurun_out/final/bench_${w}_${p}.json
  python -c "
import json,sys
d=json.load(open('gpurun_out/final/bench_${w}_${p}.json'))
print('$w $p: %.0f Msamples/s, %.0f ms/step' % (d['value'], d['ms_per_step']))"
done; done
