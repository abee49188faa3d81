#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ for THIS access pattern (every lane fetches its own random
# 128-B line with 16-B loads) on a known byte count, as MI355X_MICROARCH.md (HBM section) prescribes.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc -- $R/tools/ubench/gather_lines calib > $OUT/run.log 2>&1; echo "exit=$?"
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/calib"
fs=glob.glob(out+"/pmc/*/*_counter_collection.csv")
rows=list(csv.DictReader(open(fs[0])))
# group by dispatch id
d=collections.OrderedDict()
for r in rows:
    k=int(r["Dispatch_Id"])
    d.setdefault(k,{"name":r["Kernel_Name"][:60],"grid":int(r["Grid_Size"])})[r["Counter_Name"]]=float(r["Counter_Value"])
print("dispatch  kernel  threads  FETCH_SIZE(KB)  EA0_RDREQ  L2hit%")
res=[]
for k,v in d.items():
    h,m=v.get("TCC_HIT_sum",0),v.get("TCC_MISS_sum",0)
    res.append((k,v["name"],v["grid"],v.get("FETCH_SIZE",0),v.get("TCC_EA0_RDREQ_sum",0),100*h/max(h+m,1)))
for r in res: print(r)
open(out+"/calib.txt","w").write("\n".join(map(str,res)))
PY
find $OUT -name "*counter_collection.csv" -size +4M -delete
