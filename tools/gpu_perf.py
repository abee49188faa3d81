"""Ad-hoc GPU throughput probe (debug helper): python tools/gpu_perf.py <scene args...>"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_raytracer_amd import api

args = sys.argv[1:] or ["scenes/cornell_dragon", "-w=1200", "-s=40", "-t=10", "--seed=1"]
if args[0].startswith("scenes/cornell_dragon") and not os.path.exists("scenes/resource/dragon_high.obj"):
    import subprocess; subprocess.run(["./tools/gen_dragon", "scenes/resource/dragon_high.obj"], check=True)
t = time.time(); hs = api.HostScene(args); print("load %.2fs" % (time.time() - t), hs.log.strip(), flush=True)
t = time.time(); ds = api.DeviceScene(hs.desc, 0); print("scene_create %.2fs" % (time.time() - t), flush=True)
pipes = {"mega": api.RT_PIPELINE_MEGAKERNEL, "wf": api.RT_PIPELINE_WAVEFRONT}
sel = os.environ.get("RT_PERF_PIPES", "mega,wf").split(",")
for prec, pname, pipe in [(pr, n, pi) for (pr, n) in ((api.RT_PRECISION_F64, "f64"), (api.RT_PRECISION_F32, "f32")) for pi in sel]:
    name = pname + "-" + pipe
    for stats in (1, 0, 0):
        p = hs.params.copy(); p.precision = prec; p.collect_stats = stats; p.pipeline = pipes[pipe]
        t = time.time(); img = ds.render(hs.camera, p); wall = time.time() - t
        st = ds.stats()
        msg = f"[{name} stats={stats}] {hs.width}x{hs.height}@{hs.spp} wall {wall:.3f}s kernel {st.kernel_ms:.1f}ms (isect {st.traversal_kernel_ms:.1f}ms x{st.n_launches}) -> {st.samples/st.kernel_ms/1e3:.2f} Msamples/s mean {img[...,:3].mean():.5f} nan {int(np.isnan(img).sum())}"
        if stats:
            b = st.node_visits * st.bytes_node + st.tri_tests * st.bytes_tri
            msg += f" rays/sample {st.rays/st.samples:.2f} nodes/ray {st.node_visits/max(st.rays,1):.1f} tris/ray {st.tri_tests/max(st.rays,1):.1f} prim/ray {st.prim_tests/max(st.rays,1):.1f} alg GB/s {b/st.kernel_ms/1e6:.1f}"
        print(msg, flush=True)
