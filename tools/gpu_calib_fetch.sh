#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE for THIS access pattern (every lane fetches its own random 128-B line with
# 16-B loads) on a known byte count, as MI355X_MICROARCH.md (HBM section) prescribes.  One counter per pass.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE TCC_EA0_RDREQ_sum; do
timeout -k 10 100 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- $R/tools/ubench/gather_lines calib > $OUT/run_$c.log 2>&1; echo "$c exit=$?"
done
python3 - <<'PY'
import csv, glob, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/calib"
lines=[]
for c in ("FETCH_SIZE","TCC_EA0_RDREQ_sum"):
    fs=glob.glob(out+"/"+c+"/*/*_counter_collection.csv")
    if not fs: print(c,"missing"); continue
    for r in csv.DictReader(open(fs[0])):
        lines.append("%s dispatch %s grid %s %s = %s" % (c, r["Dispatch_Id"], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"]))
print("\n".join(lines))
open(out+"/calib.txt","w").write("\n".join(lines)+"\n")
PY
grep calib: $OUT/run_FETCH_SIZE.log
