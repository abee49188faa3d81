"""Frame pipelining on one rank's share of the headline frame (part 0 of N interleaved bands, as bench.py --gpus N partitions it):
K frames one after the other vs api.FramePipeline (frame k+1 starts under the tail of frame k).  No gather here."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rust_raytracer_amd import api
from rust_raytracer_amd import dist as rtdist
import bench

bench.ensure_dragon()
K = int(os.environ.get("PROBE_FRAMES", "6"))
hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10", "--seed=1"])
dev = torch.device("cuda", 0)
DEPTHS = [int(x) for x in os.environ.get("PROBE_DEPTHS", "1,2,3").split(",")]
pipes = {d: api.FramePipeline(hs.desc, 0, d) for d in DEPTHS}
streams = [torch.cuda.Stream(dev) for _ in range(max(DEPTHS))]
handles = [s.cuda_stream for s in streams]
base = None
for n in [int(x) for x in os.environ.get("PROBE_PARTS", "1,2,4,8").split(",")]:
    p = rtdist.partition_params(hs.params, n, 0, hs.height)
    rows = len(rtdist.rows_of_part(hs.height, n, 0))
    outs = [torch.empty((rows, hs.width, 4), dtype=torch.float64, device=dev) for _ in range(K)]
    res = {}
    for depth in DEPTHS:
        pipes[depth].render_frames(hs.camera, [p] * depth, [o.data_ptr() for o in outs[:depth]], handles)   # warm every scene
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipes[depth].render_frames(hs.camera, [p] * K, [o.data_ptr() for o in outs], handles)
        torch.cuda.synchronize()
        res[depth] = (time.perf_counter() - t0) / K
    if base is None:
        base = res
    print(f"N={n}: {rows} rows per rank; per frame " + ", ".join(f"{res[d]*1e3:.1f} ms with {d} frame(s) in flight (efficiency {base[DEPTHS[0]]/(n*res[d]) * (1 if base is res else 1):.2f})" for d in DEPTHS), flush=True)
