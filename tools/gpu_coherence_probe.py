"""How much does ray coherence matter for k_wf_mesh?  Mesh-kernel time per mesh ray for camera rays only (--max-depth=1:
neighbouring lanes = neighbouring pixels) against the full path tracer (70 % incoherent secondary rays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rust_raytracer_amd import api
import bench
bench.ensure_dragon()
for extra in (["--max-depth=1"], ["--max-depth=2"], []):
    hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10", "--seed=1"] + extra)
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy(); p.collect_stats = 1
    scene.render(hs.camera, p)
    c = scene.stats()
    p.collect_stats = 0
    scene.render(hs.camera, p)
    s = scene.stats()
    print(f"{extra}: mesh {s.traversal_kernel_ms:.1f} ms for {c.mesh_rays/1e6:.0f} M mesh rays = {s.traversal_kernel_ms*1e6/c.mesh_rays:.3f} ns per mesh ray; "
          f"nodes/mesh ray {c.node_visits/c.mesh_rays:.2f}, tris/mesh ray {c.tri_tests/c.mesh_rays:.2f}; iterations {s.n_iterations}", flush=True)
