"""A/B two builds of librt_mi355.so in separate processes."""
import subprocess, sys, os
for lib in sys.argv[1:]:
    print("=====", lib, flush=True)
    env = dict(os.environ, RT_DEVICE_LIB=lib)
    subprocess.run([sys.executable, "tools/gpu_perf.py"], env=env)
