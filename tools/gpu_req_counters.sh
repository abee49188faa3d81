#!/bin/bash
# Request counters of the wavefront kernels (one PMC pass) + the other BASELINE workloads.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/req; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*" $OUT/counters_list.txt | sort -u > $OUT/counter_names.txt
ARGS="--steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/req -- python3 $R/bench.py $ARGS > $OUT/req.log 2>&1; echo "req exit=$?"
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/ta -- python3 $R/bench.py $ARGS > $OUT/ta.log 2>&1; echo "ta exit=$?"
python3 - <<'PY'
import csv, glob, collections, os, json
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/req"
summary={}
for name in ("req","ta"):
    fs = glob.glob(f"{out}/{name}/*/*_counter_collection.csv")
    if not fs: print(name,"missing"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void rt::","")
        if "rocclr" in k: continue
        agg[k+"|"+r["Counter_Name"]] += float(r["Counter_Value"])
    summary[name]=dict(agg)
    for k,v in sorted(agg.items()):
        if ", false" in k: print(name, k, "%.6g"%v)
json.dump(summary, open(out+"/req_summary.json","w"), indent=1)
PY
find $OUT -name "*counter_collection.csv" -size +4M -delete
cd $R
for w in c2 c3; do timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench.err; echo "$w exit=$?"; cut -c1-260 $OUT/bench_$w.json; echo; done
