"""Times one rank's share of the headline frame (part 0 of N interleaved 16-row bands) for several pool sizes:
the strong-scaling efficiency a multi-GPU run can reach is bounded by this (no gather here)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rust_raytracer_amd import api
from rust_raytracer_amd import dist as rtdist
import bench
bench.ensure_dragon()
hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10", "--seed=1"])
scene = api.DeviceScene(hs.desc, 0)
dev = torch.device("cuda", 0)
def run(n, pool):
    if pool: os.environ["RT_WF_POOL"] = str(pool)
    else: os.environ.pop("RT_WF_POOL", None)
    p = rtdist.partition_params(hs.params, n, 0, hs.height)
    rows = len(rtdist.rows_of_part(hs.height, n, 0))
    out = torch.empty((rows, hs.width, 4), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    scene.render_device(hs.camera, p, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.render_device(hs.camera, p, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
base = None
for n in (1, 2, 4, 8):
    for pool in (0, 1 << 26, 1 << 25, 1 << 24):
        if n == 1 and pool: continue
        t = run(n, pool)
        if n == 1: base = t
        print(f"N={n} pool={'default' if not pool else pool>>20}M: {t*1e3:.0f} ms  -> efficiency {base/(n*t):.2f}", flush=True)
