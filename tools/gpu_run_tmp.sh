#!/bin/bash
source tools/gpu_steps.sh
step r2_tests8 900 python -m pytest tests -m gpu -q
tail -8 gpurun_out/r2_tests8.log
bash tools/gpu_sweep.sh RT_PRIM_REBUILD 0 1 -- --workload c1
