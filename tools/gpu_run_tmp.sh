#!/bin/bash
bash tools/gpu_sweep.sh RT_BVH_PAIRS 0 1
bash tools/gpu_sweep.sh RT_BVH_MAX_LEAF 2 6 8
bash tools/gpu_sweep.sh RT_WF_INNER_MIN 8 24 32
