"""What the tail of a render costs: times one rank's share of the headline frame for several partition counts and prints
the iteration count and the per-kernel sums (strong-scaling overhead = everything that does not shrink with the share)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rust_raytracer_amd import api
import bench
bench.ensure_dragon()
hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10", "--seed=1"])
scene = api.DeviceScene(hs.desc, 0)
dev = torch.device("cuda", 0)
for n in (1, 8, 16, 75):
    p = hs.params.copy()
    if n > 1:
        p.band_rows, p.n_parts, p.part = 16, n, 0
    rows = len(api.owned_rows(hs.height, p))
    out = torch.empty((rows, hs.width, 4), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.render_device(hs.camera, p, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    s = scene.stats()
    print(f"1/{n}: rows {rows}, wall {t*1e3:.1f} ms, kernels {s.kernel_ms:.1f} ms (mesh {s.traversal_kernel_ms:.1f} shade {s.shade_kernel_ms:.1f} prims {s.prims_kernel_ms:.1f}), "
          f"{s.n_iterations} iterations, samples {s.samples/1e6:.0f} M -> {s.samples/t/1e6:.0f} Msamples/s", flush=True)
