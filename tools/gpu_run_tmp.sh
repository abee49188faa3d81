#!/bin/bash
source tools/gpu_steps.sh
step final_tests 1000 python -m pytest tests -m gpu -q
tail -4 gpurun_out/final_tests.log
step final_smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
tail -2 gpurun_out/final_smoke.log
bash tools/profile.sh r02_c4_f64_final
