"""Pins the CPU oracle against the only outputs of the REAL reference that exist: the three
renders in /root/reference/samples/ (README.md:11-13), kept here as 8x8 box-filtered
statistics (tests/golden/reference_samples_ds8.npz, made by tests/golden/make_sample_fixtures.py).

The reference is entropy-seeded, so the comparison is statistical: an oracle render at 1/8
resolution (each pixel = exact box filter of radiance over 8x8 reference pixels) is compared in
LINEAR radiance with the linearised reference blocks.  One number checks geometry, camera
(incl. the defocus ring, quirk B-2), materials, the quarter-area light sampling (B-1), the
mixture pdf and the output stage at once.  CPU only; a few seconds each."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from rust_raytracer_amd import api

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_samples_ds8.npz"))


def coarse(a, k=4):
    h, w = a.shape[:2]
    return a[: h // k * k, : w // k * k].reshape(h // k, k, w // k, k, -1).mean(axis=(1, 3))


def test_light_test_matches_sample1():
    """sample1.png = scenes/light_test 2400x1600 @1000spp (README.md:12)."""
    hs = api.HostScene(["scenes/light_test", "-w=300", "-s=128", "-t=8", "--seed=1"])
    assert (hs.width, hs.height) == (300, 200) and hs.spp == 128
    img, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    lin = img[..., :3]
    ref = GOLD["sample1_linear_mean"].astype(np.float64)
    ok = GOLD["sample1_clipped_frac"] < 0.02          # blocks whose 8-bit values invert reliably
    assert ok.mean() > 0.2
    ratio = lin[ok].mean(axis=0) / ref[ok].mean(axis=0)
    # measured: 0.996 .. 0.997 per channel
    np.testing.assert_allclose(ratio, 1.0, atol=0.02)
    c_ok = coarse(GOLD["sample1_clipped_frac"][..., None])[..., 0] < 0.02
    rel = np.abs(coarse(lin) - coarse(ref)) / np.maximum(coarse(ref), 0.02)
    assert np.median(rel[c_ok]) < 0.10                # measured 0.06 (Monte-Carlo noise at 128 spp)
    # after OUR output stage the 8-bit global mean is the reference's (32.4, 26.4, 35.8) within a level or two
    tm = api.tonemap_rgb8(img).reshape(-1, 3).mean(axis=0)
    np.testing.assert_allclose(tm, GOLD["sample1_global_mean"], atol=2.0)


def test_cornell_walls_match_sample2():
    """sample2.png = scenes/cornell_dragon 1200x1200 @1000spp (README.md:13).  The dragon mesh is
    not in the checkout (.MISSING_LARGE_BLOBS); with the deterministic stand-in only regions
    dominated by walls and light are comparable."""
    obj = os.path.join(api.REPO_DIR, "scenes", "resource", "dragon_high.obj")
    if not os.path.exists(obj):
        subprocess.run([os.path.join(api.REPO_DIR, "tools", "gen_dragon"), obj], check=True)
    hs = api.HostScene(["scenes/cornell_dragon", "-w=150", "-s=128", "-t=8", "--seed=1"])
    img, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    lin = img[..., :3]
    ref = GOLD["sample2_linear_mean"].astype(np.float64)
    back = (slice(30, 45), slice(60, 90))
    np.testing.assert_allclose(lin[back].mean(axis=(0, 1)), ref[back].mean(axis=(0, 1)), rtol=0.03)   # measured 0.4 %
    green, red = (slice(60, 90), slice(5, 15)), (slice(60, 90), slice(135, 145))
    np.testing.assert_allclose(lin[green].mean(axis=(0, 1)), ref[green].mean(axis=(0, 1)), rtol=0.2)
    np.testing.assert_allclose(lin[red].mean(axis=(0, 1)), ref[red].mean(axis=(0, 1)), rtol=0.2)
    tm = api.tonemap_rgb8(img)
    assert tuple(tm[21, 75]) == (254, 254, 254)       # light patch: radiance 15 -> 254 (Appendix D)
    assert tuple(tm[1, 1]) == (0, 0, 0)               # outside the box: background black
    assert GOLD["sample2_mean"][21, 75].min() > 253.5 and GOLD["sample2_mean"][0, 0].max() < 0.5


def test_default_scene_sky_matches_sample0():
    """sample0.png = default scene 1200x800 @4000spp (README.md:11).  The sphere field is random
    (entropy seeded), the sky is not: top rows show Sky (0.2, 0.6, 2.0) through the output stage."""
    hs = api.HostScene(["-w=150", "-s=16", "--seed=1"])
    img, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    tm = api.tonemap_rgb8(img).astype(np.float64)
    ours = tm[:6].reshape(-1, 3).mean(axis=0)
    ref = GOLD["sample0_mean"][:6].reshape(-1, 3).mean(axis=0)
    np.testing.assert_allclose(ours, ref, atol=1.5)
    expect = api.tonemap_rgb8(np.array([[[0.2, 0.6, 2.0, 0.0]]]))[0, 0]
    np.testing.assert_allclose(ref, expect, atol=1.5)


def test_default_scene_metal_monkey_matches_sample0():
    """sample0.png again: the Suzanne mesh (Metal 0.8,0.6,0.2, roughness 0.05, inside a Transform) under sun + sky is at a fixed
    place; what it mirrors — the 440-sphere field — is drawn from the reference's entropy-seeded generator, so the comparison
    is a regional mean over several of OUR seeds: the monkey's blocks agree with the real Rust render to a few percent in
    linear radiance (measured per seed: 0.93 .. 1.06), the whole invertible image to ~10 % (sphere colours and count differ
    from draw to draw).  A statistical pin of Metal::scatter on mesh normals, Transform, Sun / Sky sampling with the light
    bias and the depth-of-field camera (f/2.8) against the reference itself."""
    ref = GOLD["sample0_linear_mean"].astype(np.float64)
    ok = GOLD["sample0_clipped_frac"] < 0.02
    monkey = (slice(30, 60), slice(55, 95))
    ratios, global_ratios = [], []
    for seed in (1, 2, 3):
        hs = api.HostScene(["-w=150", "-s=64", "-t=4", f"--seed={seed}"])
        assert (hs.width, hs.height) == (150, 100)
        img, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
        lin = img[..., :3]
        m = ok[monkey]
        assert m.mean() > 0.8
        ratios.append(lin[monkey][m].mean(axis=0) / ref[monkey][m].mean(axis=0))
        global_ratios.append(lin[ok].mean(axis=0) / ref[ok].mean(axis=0))
    np.testing.assert_allclose(np.mean(ratios, axis=0), 1.0, atol=0.08)
    np.testing.assert_allclose(np.mean(global_ratios, axis=0), 1.0, atol=0.15)
