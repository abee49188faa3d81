#!/bin/bash
source tools/gpu_steps.sh
step r2_tests10 900 python -m pytest tests -m gpu -q
tail -5 gpurun_out/r2_tests10.log
step partscale_r2b 600 python tools/gpu_partscale.py
grep -v amdgpu gpurun_out/partscale_r2b.log
