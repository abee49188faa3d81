"""What the tail of a render costs: one rank's share (part 0 of 8, as bench.py --gpus 8 partitions it) of the headline frame
with the per-iteration log of the wavefront driver (RT_WF_ITER_LOG=1, host check after every iteration)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_WF_ITER_LOG"] = "1"
os.environ["RT_WF_CHECK"] = "1"
import torch
from rust_raytracer_amd import api
from rust_raytracer_amd import dist as rtdist
import bench
bench.ensure_dragon()
hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10", "--seed=1"])
scene = api.DeviceScene(hs.desc, 0)
dev = torch.device("cuda", 0)
p = rtdist.partition_params(hs.params, 8, 0, hs.height)
rows = len(api.owned_rows(hs.height, p))
out = torch.empty((rows, hs.width, 4), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)
for rep in range(2):
    sys.stderr.write(f"=== render {rep}\n"); sys.stderr.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.render_device(hs.camera, p, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
s = scene.stats()
print(f"1/8 share: rows {rows}, wall {t*1e3:.1f} ms, kernels {s.kernel_ms:.1f} ms (mesh {s.traversal_kernel_ms:.1f} shade {s.shade_kernel_ms:.1f} prims {s.prims_kernel_ms:.1f}), "
      f"{s.n_iterations} iterations, {s.samples/1e6:.0f} M samples -> {s.samples/t/1e6:.0f} Msamples/s", flush=True)
