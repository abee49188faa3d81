#!/bin/bash
# PMC passes on one reduced-spp step; prints per-kernel sums.  Usage: profile_pmc.sh "<counters>" ["<counters>" ...]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for set in "$@"; do
  n=$((n+1)); rm -rf $OUT/pmc_$n
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp-divisor 10 > $OUT/pmc_$n.log 2>&1; echo "pmc $n ($set) exit=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
for d in sorted(glob.glob(f"{out}/pmc_*/")):
    fs = glob.glob(d+"*/*_counter_collection.csv")
    if not fs: print(d, "missing"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void rt::","")[:24]
        if "rocclr" in k or "advance" in k or "resolve" in k or "generate" in k: continue
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print("%-26s %-32s %.6g" % (k[0], k[1], v))
PY
find $OUT -name "*counter_collection.csv" -size +8M -delete
