"""Writes tests/golden/oracle_frames.npz: small HDR frames rendered by the CPU oracle
(oracle/liboracle.so) under fixed seeds.  The GPU parity tests compare the HIP path with these
committed frames as well as with the live oracle, so a silent change of either side shows up.

    python tests/golden/make_oracle_frames.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle  # noqa: E402
from rust_raytracer_amd import api  # noqa: E402

CASES = {
    "cornell": ["scenes/cornell", "-w=48", "-s=16", "--seed=1"],
    "light_test": ["scenes/light_test", "-w=60", "-s=16", "--seed=2"],
    "hollow_glass": ["tests/scenes/hollow_glass", "-w=48", "-s=16", "--seed=3"],
    "nested_transform": ["tests/scenes/nested_transform", "-w=48", "-s=8", "-t=2", "--seed=4"],
    "default": ["-w=60", "-s=16", "--seed=5"],
    "perlin": ["scenes/perlin", "-w=48", "-s=16", "--seed=12"],
    "texture_test": ["scenes/texture_test", "-w=48", "-s=16", "--seed=14"],
    "texture_mix": ["tests/scenes/texture_mix", "-w=48", "-s=16", "--seed=15"],
    "smoke": ["tests/scenes/smoke", "-w=48", "-s=16", "--seed=16"],
    "box_light": ["tests/scenes/box_light", "-w=48", "-s=16", "--seed=17"],
    # round 3: sphere / sky UV maps through det_acos / det_atan2, nested lights / volumes, spilled texture stacks, volumes in k_wf_prims
    "earth": ["scenes/earth", "-w=48", "-s=16", "--seed=13"],
    "sun_sky": ["tests/scenes/sun_sky", "-w=48", "-s=16", "--seed=6"],
    "cornell_smoke": ["scenes/cornell_smoke", "-w=48", "-s=16", "--seed=21"],
    "nested_volumes": ["tests/scenes/nested_volumes", "-w=48", "-s=16", "--seed=22"],
    "nested_lights": ["tests/scenes/nested_lights", "-w=48", "-s=16", "--seed=23"],
    "deep_texture": ["tests/scenes/deep_texture", "-w=48", "-s=16", "--seed=24"],
    "two_meshes": ["tests/scenes/two_meshes", "-w=48", "-s=16", "--seed=11"],
    "polished": ["tests/scenes/polished", "-w=48", "-s=16", "--seed=26"],  # Metal / Glossy with fuzz 0 (fuzzy_reflection's shortcut)
}


def main():
    out = {}
    for name, args in CASES.items():
        hs = api.HostScene(args)
        img, st = pyoracle.render(hs.desc, hs.camera, hs.params)
        out[name] = img
        print(name, img.shape, "mean", np.nanmean(img[..., :3]), "nan px", int(np.isnan(img[..., 0]).sum()))
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **out)


if __name__ == "__main__":
    main()
