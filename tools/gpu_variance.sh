#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i: %.1f Msamples/s  %.1f ms/step' % (d['value'], d['ms_per_step']))"; done
