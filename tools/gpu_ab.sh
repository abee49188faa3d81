#!/bin/bash
# Same-box A/B of device libraries:  tools/gpu_ab.sh <tag> <rounds> <workloads "c4 c2 ..."> <lib1> <lib2> ...   ("-" = the in-tree library)
# Runs the workloads with each library in turn, `rounds` times (alternating, so that clock drift hits every arm alike) and prints
# Msamples/s and the per-kernel milliseconds per step.  Output: gpurun_out/<tag>/ab_<workload>_<lib>_<round>.log, summary on stdout.
source "$(dirname "$0")/gpu_steps.sh"
tag=$1; rounds=$2; workloads=$3; shift 3
mkdir -p gpurun_out/$tag
for r in $(seq 1 $rounds); do
  for w in $workloads; do
    for lib in "$@"; do
      name=$(basename "$lib" .so); [ "$lib" == "-" ] && name=tree
      if [ "$lib" == "-" ]; then step $tag/ab_${w}_${name}_$r 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline
      else RT_DEVICE_LIB=$PWD/$lib step $tag/ab_${w}_${name}_$r 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline; fi
    done
  done
done
python3 - $tag <<'PY'
import glob, json, sys, collections
tag = sys.argv[1]
rows = collections.defaultdict(list)
for f in sorted(glob.glob(f"gpurun_out/{tag}/ab_*.log")):
    key = f.split("/ab_")[1].rsplit("_", 1)[0]
    lines = [x for x in open(f) if x.startswith("{")]
    if not lines:
        rows[key].append(None); continue
    d = json.loads(lines[0]); r = d["roofline"] or {}
    rows[key].append((d["value"], r.get("all_kernels_ms_per_step", {})))
for key, vals in rows.items():
    ok = [v for v in vals if v]
    if not ok:
        print(key, "FAILED"); continue
    ks = ok[0][1].keys()
    print(f"{key:28s} " + " / ".join(f"{v[0]:7.1f}" for v in ok) + " Msamples/s;  " + "  ".join(f"{k} " + "/".join(f"{v[1][k]:.1f}" for v in ok) for k in ks))
PY
