#!/bin/bash
# rocprofv3 passes of one bench.py configuration on the GPU box (per the guide: counters in their own passes,
# kernel trace + stats separately) and a per-kernel summary:
#     tools/profile.sh <tag> [bench.py args ...]         -> gpurun_out/<tag>/{kernel_stats.csv,pmc_summary.json,bench.json}
# Each profiled run = 1 warm-up step with counters on (`<..., true>` kernel variants) + 1 timed step.
source "$(dirname "$0")/gpu_steps.sh"
tag=$1; shift
R=$PWD; OUT=$R/gpurun_out/$tag; rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline $*"
cat rust_raytracer_amd/librt_mi355.so.srchash > "$OUT/lib_digest.txt"
timeout -k 10 600 python bench.py --steps 3 --warmup 1 $* > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
prof() { local name=$1; shift; timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1; local rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
prof kt --kernel-trace --stats
prof fetch --pmc FETCH_SIZE
prof l2 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
prof sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
prof req --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys, shutil
out = sys.argv[1]
summary = {}
f = glob.glob(f"{out}/kt/*/*kernel_stats.csv")
if f:
    shutil.copy(f[0], f"{out}/kernel_stats.csv")
    rows = list(csv.DictReader(open(f[0])))
    summary["kernel_stats"] = [{"name": r["Name"].split("(")[0].replace("void rt::", ""), "calls": int(r["Calls"]),
                                "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3,
                                "pct": float(r["Percentage"])} for r in rows]
    for k in summary["kernel_stats"][:8]:
        print(k)
for name in ("fetch", "l2", "sq", "req"):
    fs = glob.glob(f"{out}/{name}/*/*_counter_collection.csv")
    if not fs:
        print(name, "missing"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
        if "rocclr" in k:
            continue
        agg[k + "|" + r["Counter_Name"]] += float(r["Counter_Value"])
    summary[name] = dict(agg)
    for k, v in sorted(agg.items()):
        if ", false" in k or "resolve" in k:
            print(name, k, "%.6g" % v)
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
PY
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete; find "$OUT" -name "*counter_collection.csv" -size +4M -delete
