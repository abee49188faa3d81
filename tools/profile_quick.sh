#!/bin/bash
# kernel-trace stats of one reduced-spp step (spp/10) of the headline workload
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/q_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q_kt -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp-divisor 10 "$@" > $OUT/q_kt.log 2>&1; echo "kt exit=$?"
python3 - <<'PY'
import csv, glob, os
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
f = glob.glob(f"{out}/q_kt/*/*kernel_stats.csv")[0]
tot=0
rows=list(csv.DictReader(open(f)))
for r in rows: tot+=float(r["TotalDurationNs"])
for r in rows:
    print("%-28s calls %5s total %8.1f ms avg %8.1f us  %5.1f%%" % (r["Name"].split("(")[0].replace("void rt::","")[:28], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
tail -1 $OUT/q_kt.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +8M -delete
