#!/bin/bash
# Round-1 measurement script (run on the GPU box via gpurun): tests, bench line, rocprofv3 passes.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; echo "pytest exit=$?" >> $OUT/gputests.log; tail -3 $OUT/gputests.log
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit=$?"; cat $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_kt.log 2>&1; echo "rocprof kt exit=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch exit=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_l2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_l2.log 2>&1; echo "rocprof l2 exit=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/prof_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_sq.log 2>&1; echo "rocprof sq exit=$?"
find $OUT -name "*.csv" | head -40
# keep only small summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
