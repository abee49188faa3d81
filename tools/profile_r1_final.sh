#!/bin/bash
# Round-1 final measurements on the GPU box: bench line + rocprofv3 passes of the wavefront pipeline (C4, f64, full spp).
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r1final; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit=$?"; cut -c1-1500 $OUT/bench.json
timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err; echo "bench f32 exit=$?"; cut -c1-400 $OUT/bench_f32.json
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.log 2>&1; echo "kt exit=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1; echo "fetch exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 $R/bench.py $ARGS > $OUT/l2.log 2>&1; echo "l2 exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1; echo "sq exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/ic -- python3 $R/bench.py $ARGS > $OUT/ic.log 2>&1; echo "ic exit=$?"
python3 - <<'PY'
import csv, glob, collections, os, json
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r1final"
summary={}
f = glob.glob(f"{out}/kt/*/*kernel_stats.csv")
if f:
    rows=list(csv.DictReader(open(f[0]))); summary["kernel_stats"]=[{"name":r["Name"].split("(")[0].replace("void rt::",""),"calls":int(r["Calls"]),"total_ms":float(r["TotalDurationNs"])/1e6,"avg_us":float(r["AverageNs"])/1e3,"pct":float(r["Percentage"])} for r in rows]
    for k in summary["kernel_stats"][:8]: print(k)
for name in ("fetch","l2","sq","ic"):
    fs = glob.glob(f"{out}/{name}/*/*_counter_collection.csv")
    if not fs: print(name,"missing"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void rt::","")
        if "rocclr" in k: continue
        agg[k+"|"+r["Counter_Name"]] += float(r["Counter_Value"])
    summary[name]=dict(agg)
    for k,v in sorted(agg.items()):
        if ", false" in k or "resolve" in k: print(name, k, "%.6g"%v)
json.dump(summary, open(out+"/pmc_summary.json","w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
