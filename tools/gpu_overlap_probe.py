"""Do two renders that share one GPU overlap usefully (the mesh search waits on memory, the shade pass is arithmetic)?
Renders the headline frame once as a whole, then as two half frames (interleaved bands) from two host threads on two
streams, for several caps of the mesh kernel's blocks per CU (a full-occupancy persistent mesh kernel leaves no
registers for a second kernel).  Aggregate time of the pair / time of the whole frame < 1 means overlap pays."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rust_raytracer_amd import api
from rust_raytracer_amd import dist as rtdist
import bench

bench.ensure_dragon()
spp = int(os.environ.get("PROBE_SPP", "1000"))
hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", f"-s={spp}", "-t=10", "--seed=1"])
dev = torch.device("cuda", 0)
scenes = [api.DeviceScene(hs.desc, 0) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]


def render(i, n_parts, out):
    torch.cuda.set_device(0)
    p = rtdist.partition_params(hs.params, n_parts, i, hs.height)
    scenes[i].render_device(hs.camera, p, out.data_ptr(), streams[i].cuda_stream)
    streams[i].synchronize()


def whole():
    out = torch.empty((hs.height, hs.width, 4), dtype=torch.float64, device=dev)
    render(0, 1, out)
    t0 = time.perf_counter()
    render(0, 1, out)
    return time.perf_counter() - t0


def pair(concurrent):
    outs = [torch.empty((len(rtdist.rows_of_part(hs.height, 2, i)), hs.width, 4), dtype=torch.float64, device=dev) for i in range(2)]
    for rep in range(2):  # first repetition warms both pools up
        t0 = time.perf_counter()
        if concurrent:
            th = [threading.Thread(target=render, args=(i, 2, outs[i])) for i in range(2)]
            for t in th: t.start()
            for t in th: t.join()
        else:
            for i in range(2): render(i, 2, outs[i])
        dt = time.perf_counter() - t0
    return dt


base = whole()
print(f"whole frame, one render: {base*1e3:.0f} ms", flush=True)
for cap in (64, 3, 2):
    os.environ["RT_WF_MESH_BLOCKS"] = str(cap)
    w = whole()
    s = pair(False)
    c = pair(True)
    print(f"mesh blocks/CU <= {cap}: whole {w*1e3:.0f} ms | two halves back to back {s*1e3:.0f} ms | two halves concurrently {c*1e3:.0f} ms "
          f"(x{base/c:.2f} of the baseline)", flush=True)
