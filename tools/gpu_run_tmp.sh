#!/bin/bash
bash tools/gpu_sweep.sh RT_DEVICE_LIB $PWD/rust_raytracer_amd/librt_mi355.so $PWD/tools/variants_extraloads.so
