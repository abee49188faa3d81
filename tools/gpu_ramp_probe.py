"""How long does a fresh box take to reach its steady rate?  Renders the headline frame N times back to back from the first
GPU work of the process and prints wall and per-kernel milliseconds of every frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rust_raytracer_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bench.ensure_dragon()
hs = api.HostScene(bench.WORKLOADS["c4"][0])
t0 = time.time()
sc = api.DeviceScene(hs.desc, 0)
p = hs.params.copy()
p.pipeline = api.RT_PIPELINE_WAVEFRONT
print("scene on device after %.1f s" % (time.time() - t0), flush=True)
for k in range(n):
    t = time.time()
    sc.render(hs.camera, p)
    st = sc.stats()
    print("frame %2d  at %5.1f s  wall %7.1f ms   mesh %6.1f  shade %6.1f  prims %6.1f" %
          (k, time.time() - t0, (time.time() - t) * 1e3, st.traversal_kernel_ms, st.shade_kernel_ms, st.prims_kernel_ms), flush=True)
