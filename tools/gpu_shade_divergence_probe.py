"""How much of k_wf_shade's time is material divergence inside a wave?  The headline scene with (a) its own materials (lambertian
walls, glossy dragon), (b) a lambertian dragon (one scatter path for every surface), (c) glossy walls as well: shade time per ray."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_raytracer_amd import api
import bench

bench.ensure_dragon()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REPO, "scenes", "cornell_dragon")).read()
variants = {
    "as shipped (lambertian walls, glossy dragon)": src,
    "lambertian dragon": src.replace("dragon_high.obj $mat_gloss", "dragon_high.obj $mat_white"),
    "glossy walls and dragon": src.replace("$mat_white\n", "$mat_gloss\n").replace("$mat_green\n", "$mat_gloss\n").replace("$mat_red\n", "$mat_gloss\n"),
}
for name, text in variants.items():
    path = os.path.join(REPO, "scenes", "_probe_variant")
    open(path, "w").write(text)
    try:
        hs = api.HostScene([path, "-w=1200", "-s=250", "-t=10", "--seed=1"])
        scene = api.DeviceScene(hs.desc, 0)
        p = hs.params.copy()
        p.collect_stats = 1
        scene.render(hs.camera, p)
        rays = scene.stats().rays
        mesh_rays = scene.stats().mesh_rays
        p.collect_stats = 0
        scene.render(hs.camera, p)
        st = scene.stats()
        print(f"{name}: {rays/st.samples:.2f} rays per sample, {mesh_rays/rays:.2f} of them to the mesh; shade {st.shade_kernel_ms:.1f} ms = "
              f"{st.shade_kernel_ms*1e6/rays:.4f} ns per ray; prims {st.prims_kernel_ms:.1f} ms, mesh {st.traversal_kernel_ms:.1f} ms", flush=True)
        scene.close()
    finally:
        os.remove(path)
