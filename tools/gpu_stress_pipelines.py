"""Extra assurance beyond the test-suite: wavefront scheduler vs megakernel, bit for bit, over every test scene, several
seeds, image sizes and replica counts, f64; and every f32 frame finite where the f64 one is.  Prints mismatches only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from rust_raytracer_amd import api
import test_gpu_parity as T

bad = 0
n = 0
for name, args in sorted(T.SCENES.items()):
    scene_arg = [a for a in args if not a.startswith("-")]
    for seed in (101, 202, 303):
        for w, spp, t in ((37, 9, 1), (64, 32, 2), (101, 16, 1)):
            hs = api.HostScene(scene_arg + [f"-w={w}", f"-s={spp}", f"-t={t}", f"--seed={seed}"])
            sc = api.DeviceScene(hs.desc, 0)
            p = hs.params.copy()
            p.pipeline = api.RT_PIPELINE_MEGAKERNEL
            mega = sc.render(hs.camera, p)
            p.pipeline = api.RT_PIPELINE_WAVEFRONT
            wf = sc.render(hs.camera, p)
            n += 1
            same = (wf == mega) | (np.isnan(wf) & np.isnan(mega))
            if not same.all():
                bad += 1
                print(f"MISMATCH {name} seed {seed} {w}px {spp}spp t={t}: {int((~same).any(axis=2).sum())} pixels", flush=True)
            p.precision = api.RT_PRECISION_F32
            f32 = sc.render(hs.camera, p)
            if np.isnan(f32[np.isfinite(wf)]).any():
                print(f"f32 NaN where f64 is finite: {name} seed {seed} {w}px", flush=True)
print(f"{n} configurations, {bad} mismatches")
