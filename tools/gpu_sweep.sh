#!/bin/bash
# One-knob sweep / A-B of the headline bench on the GPU box:
#     tools/gpu_sweep.sh VAR v1 v2 v3 ... [-- extra bench.py args]
# runs `bench.py --steps 2 --warmup 1 --no-cpu-baseline` once per value with VAR=value in the environment
# (VAR=RT_DEVICE_LIB: A/B of differently built libraries; any RT_WF_* / RT_BVH_* knob of the library works)
# and prints Msamples/s plus the per-kernel milliseconds per step.  Output also in gpurun_out/sweep_<VAR>.log.
source "$(dirname "$0")/gpu_steps.sh"
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
    name="sweep_${var}_$(basename "$v")"
    env "$var=$v" bash -c "source tools/gpu_steps.sh; step $name 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
    python3 - "$var=$v" "gpurun_out/$name.log" <<'PY' | tee -a "gpurun_out/sweep_${var}.log"
import json, sys
tag, path = sys.argv[1:3]
lines = [x for x in open(path) if x.startswith("{")]
if not lines:
    print(tag, "FAILED"); sys.exit()
d = json.loads(lines[0]); r = d["roofline"] or {}
print(f"{tag}: {d['value']:.1f} Msamples/s, {d['ms_per_step']:.1f} ms/step, kernels/step {r.get('all_kernels_ms_per_step')}, "
      f"nodes/ray {r.get('node_visits_per_ray', 0):.2f} tris/ray {r.get('tri_tests_per_ray', 0):.2f}")
PY
done
