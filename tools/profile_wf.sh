#!/bin/bash
# rocprofv3 passes for the wavefront pipeline on C4 (f64)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wf_kt -- python3 $R/bench.py $ARGS > $OUT/wf_kt.log 2>&1; echo "kt exit=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/wf_fetch -- python3 $R/bench.py $ARGS > $OUT/wf_fetch.log 2>&1; echo "fetch exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/wf_l2 -- python3 $R/bench.py $ARGS > $OUT/wf_l2.log 2>&1; echo "l2 exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/wf_sq -- python3 $R/bench.py $ARGS > $OUT/wf_sq.log 2>&1; echo "sq exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/wf_sq2 -- python3 $R/bench.py $ARGS > $OUT/wf_sq2.log 2>&1; echo "sq2 exit=$?"
cat $OUT/wf_kt/*/*kernel_stats.csv | cut -c1-250
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
for name in ("wf_fetch","wf_l2","wf_sq","wf_sq2"):
    fs = glob.glob(f"{out}/{name}/*/*_counter_collection.csv")
    if not fs: print(name, "missing"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void rt::","")
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(name, k, f"{v:.6g}")
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
