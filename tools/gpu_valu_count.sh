#!/bin/bash
# Wave-level VALU instruction count (SQ_INSTS_VALU) of every kernel of ONE timed step of a bench configuration, per library:
#     tools/gpu_valu_count.sh <tag> "<bench args>" <lib1> <lib2> ...      ("-" = the in-tree library)
# Deterministic (same seed, same frame): separates "more instructions" from "more waiting" when two builds differ in time.
tag=$1; args=$2; shift 2
R=$PWD; mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so); [ "$lib" == "-" ] && name=tree
  out=$R/gpurun_out/$tag/valu_$name
  if [ "$lib" == "-" ]; then unset RT_DEVICE_LIB; else export RT_DEVICE_LIB=$R/$lib; fi
  timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$out" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline $args > "$out.log" 2>&1
  rc=$?; echo "valu $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 - "$out" "$name" <<'PY'
import csv, glob, collections, sys
out, name = sys.argv[1:3]
fs = glob.glob(f"{out}/*/*_counter_collection.csv")
agg = collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
    agg[k] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if "false" in k.split("<")[-1][:16] and v > 1e8:
        print(f"  {name:12s} {k:48s} SQ_INSTS_VALU {v:.4g}")
PY
  find "$out" -name "*counter_collection.csv" -size +4M -delete
done
