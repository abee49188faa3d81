#!/bin/bash
source tools/gpu_steps.sh
step r2_tests11 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x
tail -5 gpurun_out/r2_tests11.log
bash tools/gpu_sweep.sh RT_WF_DRAIN 0 4 8 16 32 64
python tools/gpu_tail_probe.py 2>&1 | tail -1
RT_WF_DRAIN=0 python tools/gpu_tail_probe.py 2>&1 | tail -1
