#!/bin/bash
# Round-end measurement on the FINAL library: rocprofv3 passes of every bench workload (tools/profile.sh), one bench line per
# workload and precision, and the default `python bench.py` (with its cpu_baseline leg).  Output under gpurun_out/r3x/.
source tools/gpu_steps.sh
mkdir -p gpurun_out/r3x/workloads
W=gpurun_out/r3x/workloads
for wl in c4 c2 c3 c1 two_meshes smoke; do
    bash tools/profile.sh r3x/${wl}_f64 --workload $wl > gpurun_out/r3x/profile_${wl}.log 2>&1 || { echo "profile $wl stopped"; exit 1; }
    cp gpurun_out/r3x/${wl}_f64/bench.json $W/${wl}_f64.json
    timeout -k 10 300 python bench.py --workload $wl --precision f32 --steps 3 --warmup 1 --no-cpu-baseline > $W/${wl}_f32.json 2> $W/${wl}_f32.err
    rc=$?; echo "bench $wl f32 rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
step r3x/bench_default 900 python bench.py
for wl in c4_3m c4_14m; do
    bash tools/profile.sh r3x/${wl} --workload $wl > gpurun_out/r3x/profile_${wl}.log 2>&1 || { echo "profile $wl stopped"; exit 1; }
    cp gpurun_out/r3x/${wl}/bench.json $W/${wl}_f64.json
done
for f in $W/*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["bound"], d["roofline"].get("frac"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -n 3 gpurun_out/r3x/bench_default.log | cut -c1-2500
