#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in "$@"; do echo "== $lib"; RT_DEVICE_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 120 python tools/gpu_wf_debug.py 2>&1 | grep "{}" | cut -c1-160; done
