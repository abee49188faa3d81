#!/bin/bash
# rocprofv3 kernel stats of the C2 (cornell, no mesh) and C3 (light_test, Suzanne) workloads.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/c2c3; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in c2 c3; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$w.log 2>&1; echo "$w exit=$?"
f=$(ls $OUT/$w/*/*kernel_stats.csv | head -1); cp $f $OUT/${w}_kernel_stats.csv; head -8 $f | cut -d, -f1-5 | cut -c1-150
done
find $OUT -name "*kernel_trace.csv" -size +2M -delete
